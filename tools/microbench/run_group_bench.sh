#!/bin/bash
# correctness of the grouping stage at edge sizes (run on the GPU box)
B=tools/microbench/group_bench
for n in 1 63 64 65 3072 3073 4095 4096 4097 8193 100000 1000003; do timeout 120 $B $n 5 1 | tail -1; done
GB=1 timeout 120 $B 300000 5 1 | tail -1
GB=8 timeout 120 $B 300000 5 1 | tail -1
GB=20 timeout 120 $B 3000000 5 1 | tail -1
GB=36 timeout 120 $B 5000000 3 1 | tail -1
RB_GROUP_T=3 timeout 120 $B 500000 5 1 | tail -1
RB_GROUP_TPB=512 timeout 120 $B 5000000 5 1 | tail -1
RB_GROUP_TPB=512 RB_GROUP_T=2 timeout 120 $B 200000 5 1 | tail -1
RB_GROUP_XCD=0 timeout 120 $B 5000000 5 1 | tail -1
timeout 300 $B 40000000 5 1 | tail -1
RB_GROUP_TPB=512 timeout 300 $B 40000000 1.2 1 | tail -1
# skewed: hot keys -> oversized buckets
timeout 300 $B 40000000 5 1 1000 | tail -1
timeout 300 $B 40000000 5 1 20 | tail -1
RB_GROUP_T=2 timeout 300 $B 2000000 5 1 50 | tail -1
GB=8 timeout 300 $B 3000000 5 1 50 | tail -1
GB=12 RB_GROUP_T=12 timeout 300 $B 30000000 5 1 50 | tail -1
