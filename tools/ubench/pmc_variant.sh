cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ONLY="$1"
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pm -- $R/tools/ubench/trunk_variants 4096 3 0 > /dev/null 2>&1
  python3 - <<'PY'
import csv,glob,collections
for f in glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    for k,v in acc.items():
        print(k, {c: round(x/3) for c,x in v.items()})
PY
done
