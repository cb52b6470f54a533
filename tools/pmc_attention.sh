#!/bin/bash
# SQ counter passes over one batch shape (default 128 x 512: the LDS-DMA attention kernel), one small counter set per pass:
#   bash tools/pmc_attention.sh <tag> [B S] -> gpurun_out/<tag>/attn_pmc_<B>_<S>.json (per kernel: counter sums over 3 forwards)
TAG=${1:-r03}; B=${2:-128}; S=${3:-512}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/pmc_list_avail.txt 2>&1 || true
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_WAVE32_LDS"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/attn_pmc_${B}_${S}_p$i -- python3 $R/tools/one_shape.py $B $S 3 > /dev/null 2> $OUT/attn_pmc_p$i.err || echo "pass $i ($set) failed: $(tail -n 2 $OUT/attn_pmc_p$i.err | cut -c1-300)"
done
python3 - <<PY
import collections, csv, glob, json
agg = collections.OrderedDict()
for f in glob.glob("$OUT/attn_pmc_${B}_${S}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_attn" not in name and "k_proj" not in name:
            continue
        agg.setdefault(name, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
json.dump({k: dict(v) for k, v in agg.items()}, open("$OUT/attn_pmc_${B}_${S}.json", "w"), indent=1)
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:34s} {x:.4g}")
PY
