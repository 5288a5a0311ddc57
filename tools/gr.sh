#!/bin/bash
# usage: gr.sh <timeout> <command string>; retries while the pod has no free GPU slot (exit 3)
T=$1; shift
for i in $(seq 1 40); do
  gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
