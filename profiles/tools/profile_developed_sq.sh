# SQ counters of the fused-stage kernels in ambient gas (the headline's first steps) and in developed flow (bench.py --developed-only), same box:
#   bash profiles/tools/profile_developed_sq.sh TAG   -> gpurun_out/TAG/{ambient,developed}_sq{1,2}.txt
tag=${1:-dev_sq}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
rocprofv3 -L 2>/dev/null | grep -oE "SQ_INSTS_[A-Z_0-9]+|SQ_[A-Z_]*BRANCH[A-Z_]*" | sort -u > $O/avail_sq_insts.txt
run() { # name, extra flag
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/$1_1 -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-secondary $2 > $O/$1_1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_CBRANCH SQ_INSTS_CBRANCH_TAKEN -d $O/$1_2 -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-secondary $2 > $O/$1_2.log 2>&1
  for d in 1 2; do f=$(find $O/$1_$d -name "*.db" | head -1); [ -n "$f" ] && python profiles/summarize_rocpd.py "$f" > $O/$1_sq$d.txt 2>&1; done
  find $O -name "*.db" -delete; rm -rf $O/$1_1 $O/$1_2
}
run ambient ""
run developed "--developed-only"
ls $O; head -40 $O/developed_sq1.txt | cut -c1-200
