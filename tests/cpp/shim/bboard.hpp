// Stand-in for pomcpp's include/bboard.hpp when existing sources are compiled against this repo: the whole
// surface is include/pom_bboard.hpp.
#include "pom_bboard.hpp"
