cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a/sq
mkdir -p $O
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/sq1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $O/sq2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/sq2.log 2>&1
for d in sq1 sq2; do
  f=$(find $O/$d -name "*.db" | head -1)
  if [ -n "$f" ]; then python profiles/summarize_rocpd.py "$f" > $O/$d.txt 2>&1; fi
done
find $O -name "*.db" -delete
tail -3 $O/sq1.log | cut -c1-300
grep -A60 "counters" $O/sq1.txt | grep -E "sweep|pre" | cut -c1-60,100-170
