#!/bin/bash
# N ranks of one host-mirror executable on ONE GPU through the shm test transport (the ranks must share a parent process: its pid is part of the
# rendezvous tag); per-rank logs under $OUT; a rank waits TMO seconds for a peer.  usage: run_cxx_ranks.sh N OUT exe args...
N=$1; OUT=$2; EXE=$3; shift 3
mkdir -p $OUT
pids=()
for r in $(seq 0 $((N-1))); do
  RANK=$r WORLD_SIZE=$N LOCAL_RANK=$r MASTER_PORT=${PORT:-29901} MASTER_ADDR=127.0.0.1 QK_COMM_BACKEND=shm HSA_ENABLE_IPC_MODE_LEGACY=0 \
    QK_COMM_TIMEOUT=${TMO:-60} $EXE "$@" > $OUT/rank$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
for r in $(seq 0 $((N-1))); do echo "== rank $r"; tail -${TAIL:-25} $OUT/rank$r.log; done
exit $rc
