#!/bin/bash
# the reference's unchanged RadhydroShell file on the C++ host, deck of BASELINE config 4, N steps: the executable's own figure of merit
# usage: bash profiles/tools/cxx_shell_run.sh [steps] [extra deck overrides ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
steps=${1:-22}; shift
tmp=$(mktemp -d); cp $R/tests/golden/dust_shell_initial_conditions.txt $tmp/initial_conditions.txt; cd $tmp
$R/quokka_amd/host/bin/ref_RadhydroShell $R/quokka_amd/host/decks/radhydro_shell_256.in max_timesteps=$steps plotfile_interval=-1 checkpoint_interval=-1 hydro.rk2_carry_rhs=1 "$@" 2>&1 | grep -E "figure-of-merit|elapsed time|abort" | head -4
rm -rf $tmp
