# Collect SQ/TA/TCP counters for the hot kernels (separate rocprofv3 --pmc passes, kernel trace only).
# usage (through gpurun): bash tools/pmc_sq.sh [extra env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sq
mkdir -p $OUT
pass() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$n -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-leg > $OUT/pass$n.log 2>&1
  f=$(find /tmp/pmc_$n -name '*counter_collection.csv' | head -1)
  python - "$f" "$@" > $OUT/pass$n.txt <<'PY'
import csv,sys,re
from collections import defaultdict
agg=defaultdict(lambda: defaultdict(float)); cnt=defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    k=k.replace("(anonymous namespace)::","")
    k=re.sub(r"\(.*","",k).replace("void ","")
    if "rocprim" in k: k="rocprim:"+k.split("::")[-1][:40]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    cnt[(k,r["Counter_Name"])]+=1
names=sys.argv[2:]
print("kernel,dispatches,"+",".join(names))
for k,v in sorted(agg.items(), key=lambda kv:-max(kv[1].values())):
    print(k+","+str(cnt[(k,names[0])])+","+",".join("%.4g"%v.get(n,0) for n in names))
PY
}
pass 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pass 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC
pass 3 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH
pass 4 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES
head -14 $OUT/pass1.txt $OUT/pass2.txt
