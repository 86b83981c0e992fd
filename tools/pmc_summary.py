"""Fold rocprofv3 outputs into the per-launch HBM-traffic record bench.py reads (profiles/rNN_pmc_hbm_traffic_per_launch.json).

Collect (each counter in its OWN pass, with --kernel-trace only -- MI355X_MICROARCH.md, HBM / rocprofv3 section):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --no-kernel-timing
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --no-kernel-timing
then
  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_hbm_traffic_per_launch.json

Units / corrections: FETCH_SIZE and WRITE_SIZE count KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests as
64 bytes, so reads are doubled (fetch_MB_corrected).  Values are averages per launch of each kernel."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r'([A-Za-z_][A-Za-z0-9_]*)(<[^(]*>)?\(', name)
    if not m:
        return name.strip()
    return (m.group(1) + (m.group(2) or '')).strip()


def fold(directory, counter):
    agg = {}
    files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    assert files, f'no *counter_collection.csv under {directory}'
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            k = short(row['Kernel_Name'])
            d = agg.setdefault(k, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += float(row['Counter_Value'])
            if row.get('End_Timestamp') and row.get('Start_Timestamp'):
                d[2] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-6
    return agg


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fe, wr = fold(fetch_dir, 'FETCH_SIZE'), fold(write_dir, 'WRITE_SIZE')
    rec = {'_note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1` (b=8, 1024^2); '
                    'KiB -> MB; fetch_MB_corrected = 2 x raw (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM '
                    'section); per-launch averages; avg_ms from the dispatch timestamps of the FETCH pass (tools/pmc_summary.py)'}
    rows = sorted(fe.items(), key=lambda kv: -kv[1][2])
    for k, (n, kib, ms) in rows[:24]:
        w = wr.get(k, [1, 0.0, 0.0])
        rec[k] = dict(calls=n, avg_ms=round(ms / n, 4), fetch_MB_raw=round(kib * 1024 / 1e6 / n, 2),
                      fetch_MB_corrected=round(2 * kib * 1024 / 1e6 / n, 2), write_MB=round(w[1] * 1024 / 1e6 / max(w[0], 1), 2))
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
