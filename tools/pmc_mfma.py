"""Fold a rocprofv3 SQ counter pass (SQ_INSTS_VALU, SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES,
GRBM_GUI_ACTIVE over `bench.py --steps 1 --warmup 1`) into per-kernel MFMA utilisation figures (profiles/rNN_mfma_busy.json).

  python tools/pmc_mfma.py <counter dir> <out.json>

Per kernel (averages per launch over all launches of the pass):
  mfma_share_of_vector_issue = 64 N_mfma / (64 N_mfma + 4 N_valu)      v_mfma_f32_32x32x2_f32 holds the SIMD's vector pipe for 64
      cycles, another vector instruction for ~4 (MI355X_MICROARCH.md, cycle constants): the share of the vector pipe's time that
      goes into matrix instructions -- what DESIGN.md §4 uses to explain the fp32-MFMA kernels' distance from peak;
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)   matrix-pipe busy cycles per SIMD-cycle of the
      dispatch (256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs); reported as measured, the counter's unit on gfx950
      is taken from the guide (cycles), so read it as a relative figure between kernels first.
N_valu in SQ_INSTS_VALU includes the MFMA instructions on this chip generation; both readings are given."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'([A-Za-z_][A-Za-z0-9_]*)(<[^(]*>)?\(', name)
    return ((m.group(1) + (m.group(2) or '')).replace(' ', '')) if m else name.strip()


def main():
    d, out = sys.argv[1:3]
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(float)
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    assert files, f'no *counter_collection.csv under {d}'
    for f in files:
        for row in csv.DictReader(open(f)):
            k = short(row['Kernel_Name'])
            c = row['Counter_Name']
            agg[k][c] += float(row['Counter_Value'])
            cnt[k][c] += 1
            if c == 'SQ_WAVE_CYCLES' and row.get('End_Timestamp') and row.get('Start_Timestamp'):
                dur[k] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-6
    rec = {'_note': __doc__.split('\n\n')[0] + ' Formulas: see tools/pmc_mfma.py.'}
    rows = []
    for k, cs in agg.items():
        n = max(cnt[k].values())
        v = {c: cs[c] / cnt[k][c] for c in cs}
        mf, va = v.get('SQ_INSTS_MFMA', 0.0), v.get('SQ_INSTS_VALU', 0.0)
        if mf <= 0:
            continue
        other_incl = max(va, 0.0)               # reading A: SQ_INSTS_VALU counts the non-matrix vector instructions only
        other_excl = max(va - mf, 0.0)          # reading B: it includes the matrix instructions
        gui = v.get('GRBM_GUI_ACTIVE', 0.0)
        r = dict(launches=n, ms_total=round(dur[k], 3), insts_mfma=round(mf), insts_valu=round(va),
                 valu_per_mfma=round(va / mf, 3),
                 mfma_share_of_vector_issue=round(64 * mf / (64 * mf + 4 * other_incl), 4),
                 mfma_share_of_vector_issue_if_valu_includes_mfma=round(64 * mf / (64 * mf + 4 * other_excl), 4),
                 mfma_busy_cycles=round(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)), sq_busy_cycles=round(v.get('SQ_BUSY_CYCLES', 0.0)),
                 sq_wave_cycles=round(v.get('SQ_WAVE_CYCLES', 0.0)), grbm_gui_active=round(gui),
                 mfma_busy_frac=round(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui / 8.0 * 1024.0), 4) if gui > 0 else None,
                 # effective shader clock of the dispatch (guide, DVFS give-back: GRBM_GUI_ACTIVE / 8 / wall time; reads high on short ones)
                 eff_clock_ghz=round(gui / 8.0 / (dur[k] / n * 1e6), 3) if gui > 0 and dur[k] > 0 else None)
        rows.append((dur[k], k, r))
    for _, k, r in sorted(rows, reverse=True):
        rec[k] = r
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
