# GPU idle gaps of the C++ host on BASELINE config 5 and what the host was doing: -> gpurun_out/<tag>/gaps.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-cxx_amr_gaps}; mkdir -p $O; B=$R/quokka_amd/host
(cd $B && rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $O/ht -- $B/bin/ref_HydroBlast3D $B/decks/blast_amr_maxlev2.in max_timesteps=55 hydro.rk2_carry_rhs=1 plotfile_interval=-1 checkpoint_interval=-1 > $O/ht.log 2>&1)
python $R/profiles/tools/gpu_gaps.py $O/ht 15 > $O/gaps.txt 2>&1
rm -rf $O/ht
grep figure-of-merit $O/ht.log; cat $O/gaps.txt
