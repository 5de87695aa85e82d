# GPU idle gaps of any command (profiles/tools/gpu_gaps.py): bash profiles/tools/profile_gaps.sh TAG [MIN_US] -- command ...   -> gpurun_out/TAG/gaps.txt
tag=$1; shift; min=15; if [ "$1" != "--" ]; then min=$1; shift; fi; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
(cd $R && rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $O/ht -- "$@" > $O/run.log 2>&1)
python $R/profiles/tools/gpu_gaps.py $O/ht $min > $O/gaps.txt 2>&1
rm -rf $O/ht
tail -1 $O/run.log | cut -c1-160; head -24 $O/gaps.txt
