#!/bin/bash
# MFMA-busy fraction of hs_gemm_nt's 256 x 256 tile on the stage-2 Mlp products (HEAL-SWIN-B @ nside 256, batch 8): one rocprofv3 --pmc pass
# (with --kernel-trace only) per epilogue.  Counters are summed over their instances: SQ_VALU_MFMA_BUSY_CYCLES over the chip (= 32 cycles x MFMA instructions),
# GRBM_GUI_ACTIVE over the 8 XCDs; mfma_busy = busy cycles / (1024 SIMDs x active cycles per XCD) = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE).
# usage: bash tools/collect_gemm_pmc.sh OUT.txt
OUT=${1:-gpurun_out/gemm_pmc.txt}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
: > $ROOT/$OUT
for spec in "98304 2048 512 3 0 bias" "98304 2048 512 3 1 gelu" "98304 2048 512 3 2 dgelu" "98304 512 2048 3 0 fc2_bias"; do
  set -- $spec
  rm -rf /tmp/pmc_gemm
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/pmc_gemm -o t -- python $ROOT/tools/bench_gemm_one.py $1 $2 $3 $4 $5 6 > /dev/null 2>&1
  python - "$6" $(find /tmp/pmc_gemm -name '*counter_collection.csv' | head -1) >> $ROOT/$OUT <<'PY'
import csv, sys, collections
tag, path = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    if "gemm_nt_kernel" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = [{k: sum(v) for k, v in d.items()} for d in acc.values()][1:]  # (first launch: cold)
n = len(rows)
m = {k: sum(r[k] for r in rows) / n for k in rows[0]}
print(f"{tag:10s} launches {n}: mfma_busy = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 128 / m['GRBM_GUI_ACTIVE']:.3f}  (SQ_VALU_MFMA_BUSY_CYCLES {m['SQ_VALU_MFMA_BUSY_CYCLES']:.0f}, GRBM_GUI_ACTIVE {m['GRBM_GUI_ACTIVE']:.0f}, "
      f"MFMA insts {m['SQ_INSTS_MFMA']:.0f}, VALU insts per MFMA {m['SQ_INSTS_VALU'] / m['SQ_INSTS_MFMA']:.2f})")
PY
done
cat $ROOT/$OUT
