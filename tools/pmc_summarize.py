"""Turn the counter CSVs of tools/gpu_pmc.sh (gpurun_out/pmc_<tag>/p*/p_counter_collection.csv) into the per-launch summary
bench.py reads for roofline.traffic.  usage: pmc_summarize.py <tag> <out.json> [workload text] [algorithmic bytes]"""
import csv, collections, glob, json, sys
tag, out = sys.argv[1], sys.argv[2]
agg, names = collections.defaultdict(list), collections.Counter()
for p in sorted(glob.glob(f'gpurun_out/pmc_{tag}/p*/p_counter_collection.csv')):
    for r in csv.DictReader(open(p)):
        n = r['Kernel_Name']
        if 'conv_' in n and 'reduce' not in n and 'pack' not in n:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            names[n.split('(')[0].replace('void ', '').strip()] += 1
c = {k: sum(v) / len(v) for k, v in agg.items()}
fetch_raw, write = c['FETCH_SIZE'] * 1024, c['WRITE_SIZE'] * 1024            # counters are in KB
res = {
    'kernel': names.most_common(1)[0][0],          # as rocprofv3 prints it; bench.py only uses the traffic when this matches its dispatch
    'workload': sys.argv[3] if len(sys.argv) > 3 else '3x3 256->256 @ 8x128x128 bf16 (tools/conv_only.py fwd)',
    'counters_mean_per_launch': c,
    'fetch_bytes_raw': fetch_raw, 'fetch_bytes_corrected_x2': 2 * fetch_raw, 'write_bytes': write,
    'traffic_bytes': 2 * fetch_raw + write,
    'algorithmic_bytes': int(sys.argv[4]) if len(sys.argv) > 4 else 2 * 8 * 128 * 128 * 256 * 2 + 256 * 2304 * 2,
    'note': 'FETCH_SIZE / WRITE_SIZE in KB from separate rocprofv3 --pmc passes; gfx950 reports 1/2 of wide coalesced reads '
            '(MI355X_MICROARCH.md, HBM) -> x2 on the read side',
}
if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
    # both are chip-wide sums: MFMA-busy cycles over 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs -> busy fraction per SIMD = ratio * 8 / 1024
    res['mfma_util'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] * 128)
    res['effective_clock_ghz_hint'] = 'GRBM_GUI_ACTIVE / 8 / kernel duration (see the durations line of the collection log)'
json.dump(res, open(out, 'w'), indent=1)
print(res['kernel'], 'traffic MB', res['traffic_bytes'] / 1e6, 'mfma_util', res.get('mfma_util'))
