#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof_head -o head -- python3 $R/tools/head_graph_probe.py 8 > $O/probe.txt 2>&1
ls $O/prof_head | head
