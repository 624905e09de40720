#!/bin/bash
# host analysis: flood-fill level statistics of the device policy body (see tests/emul/flood_levels.cpp); runs anywhere
set -e
cd "$(dirname "$0")/../.."
mkdir -p build
INC="-Iinclude -Ipomcpp_amd/csrc -Ioracle"
g++ -O2 -std=c++17 -Wno-unknown-pragmas -DPOM_LEVEL_STATS $INC -c tests/emul/pom_policy_emul.cpp -o build/pom_policy_emul_stats.o
gcc -O2 -std=c11 $INC -c oracle/pom_oracle.c -o build/fl_oracle.o
gcc -O2 -std=c11 $INC -c oracle/pom_policy_oracle.c -o build/fl_policy.o
gcc -O2 -std=c11 $INC -c oracle/pom_boardgen_oracle.c -o build/fl_boardgen.o
g++ -O2 -std=c++17 $INC tests/emul/flood_levels.cpp build/pom_policy_emul_stats.o build/fl_oracle.o build/fl_policy.o build/fl_boardgen.o -o build/flood_levels
./build/flood_levels "${1:-64}" "${2:-400}"
