#!/bin/bash
cd "$(dirname "$0")/.."
g=gpurun_out/r06e
run() { name=$1; shift; env "$@" python tools/dbg_graph.py $name > ${g}_dbg_$name.txt 2>&1; grep "^$name" ${g}_dbg_$name.txt | cut -c1-200; }
run fillkernel X=1
run memset ADAMVS_ZERO_FILL_KERNEL=0
run fillkernel_noeager X=1
run memset_noeager ADAMVS_ZERO_FILL_KERNEL=0
run hostkernarg HIP_FORCE_DEV_KERNARG=0
run hostkernarg_memset HIP_FORCE_DEV_KERNARG=0 ADAMVS_ZERO_FILL_KERNEL=0
