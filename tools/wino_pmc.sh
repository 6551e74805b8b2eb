cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/wpmc
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/wpmc -o w -- python3 $R/tools/${PMC_PROG:-wino_pmc.py} > $R/gpurun_out/wpmc.log 2>&1
echo rc=$?
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('$R/gpurun_out/wpmc/w_counter_collection.csv')):
    if "wino" in r["Kernel_Name"]:
        agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k)
    for c, vals in v.items():
        print('   %-28s %14.0f (n=%d)' % (c, sum(vals[2:]) / max(1, len(vals[2:])), len(vals)))
PY
rm -rf $R/gpurun_out/wpmc
