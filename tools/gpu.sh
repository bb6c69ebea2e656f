#!/bin/bash
# gpurun front end: waits for a free box (exit code 3 = none free, nothing ran, nothing charged) and then
# runs the command ONCE.  usage: tools/gpu.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > /tmp/gpu_sh.out 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" /tmp/gpu_sh.out; then cat /tmp/gpu_sh.out | tail -${TAILN:-15}; exit $rc; fi
  sleep 30
done
echo "no box after 40 tries"; tail -3 /tmp/gpu_sh.out; exit 3
