cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for only in "no fragment reads" "no staging" "x16<128,4,1> 16x16x32"; do
  export ONLY="$only"
  rm -rf /tmp/pm; rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pm -- $R/tools/ubench/trunk_variants 4096 3 0 > /dev/null 2>&1
  python3 - <<'PY'
import csv,glob,collections,os
for f in glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:70]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items():
        if 'trunk' in k: print(os.environ['ONLY'], '|', k, {c: round(x/3) for c,x in v.items()})
PY
done
