#!/bin/bash
# tools/parity_at_size.py over the shapes the kernels of rounds 3-4 changed (run through gpurun): every engine against the CPU oracle, bit for bit
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06}_parity_at_size.txt
echo "# tools/parity_at_size.py on one MI355X (round-6 tree, csrc id $(python tools/csrc_id.py): conflict path on cached table slots with one-kernel labelling rounds — single-GPU and sharded —, closed experiments pruned from the insert path, packed-host ingest; the swept Bloom-bit stage forced in two runs, the walkers cross-checked in the last): single-GPU engine and sharded engine (8 virtual ranks split reads, 2 replicated hashing; both drivers) against the CPU oracle, bit for bit" > $O
run() { echo "## $1" >> $O; shift; timeout 1500 env "$@" 2>&1 | grep -v "amdgpu.ids" >> $O; }
run k25_16M A=1 python tools/parity_at_size.py 16000000 25 72000000
run k35_8M A=1 python tools/parity_at_size.py 8000000 35 36000000
run k47_4M A=1 python tools/parity_at_size.py 4000000 47 18000000
run k63_4M A=1 python tools/parity_at_size.py 4000000 63 18000000
run k28_4M A=1 python tools/parity_at_size.py 4000000 28 18000000
run k25_8M_errors_unmasked RB_SYNTH_KEEP_ERRORS=1 python tools/parity_at_size.py 8000000 25 36000000
run k25_8M_errors_unmasked_swept RB_SYNTH_KEEP_ERRORS=1 RB_SWEEP=1 python tools/parity_at_size.py 8000000 25 36000000
run k35_8M_swept A=1 RB_SWEEP=1 RB_PF_SKIP=2 python tools/parity_at_size.py 8000000 35 36000000
run k25_8M_walkers_cross_checked A=1 RB_FILTER_CHECK=1 python tools/parity_at_size.py 8000000 25 36000000
