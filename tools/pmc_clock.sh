# effective shader clock during the L2 kernels: GRBM_GUI_ACTIVE cycles / kernel duration
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
rm -rf /tmp/pc; R3DM_L2_INT_VARIANT=${1:-2} timeout 90 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pc -- python tools/gpu_perf.py --images 16 --feat 8192 --reps 0 --integer-mfma > /tmp/pc.log 2>&1
python - <<'PY' | tee gpurun_out/pmc_clock.txt
import csv, glob
cc = glob.glob('/tmp/pc/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob('/tmp/pc/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    if 'l2_knn2' in r['Kernel_Name']:
        dur[r['Dispatch_Id']] = (r['Kernel_Name'][:60], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for r in csv.DictReader(open(cc)):
    if r['Dispatch_Id'] in dur and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        n, d = dur[r['Dispatch_Id']]
        print(n, 'cycles', r['Counter_Value'], 'ns', d, 'GHz %.3f' % (float(r['Counter_Value']) / d))
PY
