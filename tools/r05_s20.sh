#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CM_HOST_MARKS=1 python tools/lone_loop.py > gpurun_out/r05y_marks.txt 2>&1
grep -n "setup" gpurun_out/r05y_marks.txt | tail -3
awk '/\[host\] setup/{buf=""} {buf=buf"\n"$0} END{print buf}' gpurun_out/r05y_marks.txt | head -80
