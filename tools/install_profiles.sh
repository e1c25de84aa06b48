#!/bin/bash
# copy the outputs of tools/profiles.sh <tag> from gpurun_out/ into profiles/ (names carry the round) and rebuild traffic.json
T=${1:?tag}; G=gpurun_out
cp $G/bench_$T.json profiles/${T}_bench_line_default.json
cp $G/bench_${T}_detail.json profiles/${T}_bench_detail_default.json
cp $G/bench_${T}_cfg2.json profiles/${T}_bench_line_under_rocprofv3_cfg2.json
cp $G/bench_${T}_cfg3.json profiles/${T}_bench_line_cfg3_65536rays.json
cp $G/bench_${T}_cfg3_8192.json profiles/${T}_bench_line_cfg3_8192rays.json
cp $G/bench_${T}_cfg4.json profiles/${T}_bench_line_cfg4.json
cp $G/bench_${T}_extract.json profiles/${T}_bench_line_extract_512cubed_production_tile.json
cp $G/bench_${T}_extract_cfg2.json profiles/${T}_bench_line_extract_512cubed_cfg2_fields.json
cp $G/kernel_stats_${T}_cfg3_8192.txt profiles/${T}_kernel_stats_cfg3_8192rays_bench_steps5_warmup2.txt
cp $G/pmc_summary_${T}_cfg3.txt profiles/${T}_pmc_summary_cfg3.txt
cp $G/pmc_summary_${T}_cfg4.txt profiles/${T}_pmc_summary_cfg4.txt
cp $G/kernel_stats_${T}_cfg2.txt profiles/${T}_kernel_stats_cfg2_bench_steps10_warmup3.txt
cp $G/kernel_stats_${T}_cfg3.txt profiles/${T}_kernel_stats_cfg3_bench_steps4_warmup2.txt
cp $G/kernel_stats_${T}_cfg4.txt profiles/${T}_kernel_stats_cfg4_bench_steps4_warmup2.txt
cp $G/kernel_gaps_${T}_cfg2.txt profiles/${T}_kernel_gaps_cfg2.txt
cp $G/pmc_summary_$T.txt profiles/${T}_pmc_summary.txt
[ -f $G/soak_$T.txt ] && grep -v amdgpu.ids $G/soak_$T.txt > profiles/${T}_soak_1000_iterations.txt
python tools/pmc_traffic.py profiles/${T}_pmc_summary.txt profiles/traffic.json > /dev/null
python - <<PY
import json
d = json.load(open('profiles/${T}_bench_detail_default.json'))
print('cfg2', round(d['ms_per_step'], 2), 'ms', round(d['value'] / 1e6, 3), 'M rays/s; reference schedule', round(d['value_reference_schedule'] / 1e6, 2))
r = d['roofline']; print(' roofline', r['kernel'], round(r['frac'], 3), round(r['avg_launch_ms'], 3), 'ms', 'traffic', r['traffic'])
for r in d['roofline_kernels']: print('  ', r['kernel'][:60], r['avg_launch_ms'], r['frac'])
print(' cpu', round(d['cpu_baseline']['value']), 'e2e', round(d['end_to_end']['frac_of_binding'], 3))
for k, v in d.get('secondary', {}).items(): print(' ', k, round(v.get('ms_per_step', 0), 2), 'ms', round(v.get('value', 0)), v.get('frac_of_binding'))
PY
