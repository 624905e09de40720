// Stand-in for pomcpp's include/step_utility.hpp: the helpers agents call live in include/pom_bboard.hpp.
#include "pom_bboard.hpp"
