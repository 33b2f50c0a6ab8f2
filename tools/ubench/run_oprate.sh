#!/bin/bash
# one process per kernel (a kernel that does not come back only costs its own 20 s)
cd "$(dirname "$0")"
N=$(grep -c "__global__" oprate.hip)
for i in $(seq ${1:-0} $((N - 1))); do
  timeout 20 ./oprate $i || echo "kernel $i: no result (rc $?)"
done
