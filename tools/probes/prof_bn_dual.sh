cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PFST_WGRAD_STREAM=0 PFST_FORK_TEACHER=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bn -o s -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-math --no-kernel-timing > gpurun_out/s3_bench.json 2> gpurun_out/s3_bench.err
find gpurun_out/prof_bn -name "*kernel_stats.csv" -exec cp {} gpurun_out/s3_kernel_stats.csv \;
find gpurun_out/prof_bn -name "*kernel_trace.csv" -exec cp {} gpurun_out/s3_kernel_trace.csv \;
rm -rf gpurun_out/prof_bn
grep -i "bn_bwd" gpurun_out/s3_kernel_stats.csv | cut -c1-60,400-
python3 - <<'P'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/s3_kernel_trace.csv')) if 'bn_bwd' in r['Kernel_Name'] and 'dual' in r['Kernel_Name']]
for r in rows[-16:]:
    print(r['Kernel_Name'][:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else '', r.get('Grid_Size_Y',''), r.get('Grid_Size_Z',''), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
P
rm -f gpurun_out/s3_kernel_trace.csv
