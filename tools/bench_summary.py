"""Print the numbers of a default bench line (and optionally a stripes line) one per row: python tools/bench_summary.py <bench.json> [<stripes.json>]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('C3 ms', round(d['ms_per_step'], 3), 'Mpts/s', round(d['value'] / 1e6, 1), 'generic', round(d['generic_durations']['ms_per_step'], 3), 'sustained', round(d['sustained']['ms_per_step'], 3),
      'full_callback', round(d['full_callback_ms'], 3), 'perturbed', round(d['full_callback_perturbed_ms'], 3))
r = d['roofline']
print('  k_solve merged/serialized', round(r['kernel_ms_per_step'], 3), round(r['kernel_ms_serialized_per_step'], 3), 'k_round merged/serialized', round(r['k_round_ms_per_step'], 3), round(r['k_round_ms_serialized_per_step'], 3),
      'device serialized', round(r['device_ms_serialized_per_step'], 3), 'launches', r['launches_per_step'])
print('  roofline hbm achieved GB/s', round(r['achieved'], 3), 'frac', r['frac'], 'traffic', r['traffic'], 'traffic_total', r.get('traffic_total'))
f = r['fp64']
print('  fp64 frac k_solve', round(f['k_solve']['frac'], 4), 'serialized', round(f['k_solve_serialized']['frac'], 4), 'k_round', round(f['k_round']['frac'], 4), 'serialized', round(f['k_round_serialized']['frac'], 4),
      'whole', round(f['whole_evaluation']['frac'], 4), 'gflop', round(f['whole_evaluation']['gflop'], 2), 'k_round table evals', f['k_round_table_evals_per_step'])
n = d['north_star']
print('NS ms', round(n['ms_per_step'], 3), 'generic', round(n['generic_durations']['ms_per_step'], 3), 'k_solve', round(n['k_solve_ms_per_step'], 3), 'k_round', round(n['k_round_ms_per_step'], 3),
      'cpu', round(n['cpu_baseline']['value']), 'x', round(n['speedup_vs_cpu_baseline']))
m = d['map_distribution']; print('map ms', round(m['ms_per_step'], 3), 'generic', round(m['generic_durations']['ms_per_step'], 3))
for k, v in d['other_configs'].items(): print(k, 'ms', round(v['ms_per_step'], 3), 'generic', round(v['generic_durations_ms_per_step'], 3), v['plan'])
c = d['c4_one_gpu']; print('C4 one GPU ms', round(c['ms_per_step'], 3), 'Mpts/s', round(c['value'] / 1e6, 1), c['plan'])
fo = d['first_optimisation']; print('first optimisation total', round(fo['total_ms'], 1), 'first', round(fo['first_callback_ms'], 2), 'steady', round(fo['steady_callback_ms'], 3), 'settled after', fo['callbacks_until_plan_settled'])
for k, v in d['reference_scale']['cases'].items():
    print('ref', k, 'median', round(v['callback_us_median'], 1), 'p10', round(v['callback_us_p10'], 1), 'p90', round(v['callback_us_p90'], 1), 'device', round(v['device_us'], 1), 'oracle12 ms', round(v['oracle_ms_threads_12'], 2),
          'x', round(v['speedup_vs_oracle_threads_12'], 1), 'cost err', v['cost_rel_err_vs_oracle'], 'grad err', v['grad_rel_err_vs_oracle'], v['plan']['gsip_bound_mode'])
s = d['sustained']; print('sustained median', round(s['median_ms'], 3), 'p99', round(s['p99_ms'], 3), 'max', round(s['max_ms'], 3), 'last/first', round(s['last_over_first'], 4), 'clock', round(s['shader_clock_mhz_first']), round(s['shader_clock_mhz_last']))
print('cpu baseline', round(d['cpu_baseline']['value']), 'x', round(d['speedup_vs_cpu_baseline']), d['cpu_baseline']['cores'], 'threads')
if len(sys.argv) > 2:
    s = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    st = s['stripes']
    print('stripes: max', round(st['device_ms_max'], 3), 'mean', round(st['device_ms_mean'], 3), 'max/mean', round(st['device_ms_max_over_mean'], 4), 'fixed host us', round(st['fixed_host_us_per_evaluation'], 1),
          'ideal on 8', round(st['ideal_ms_per_step_on_8_gpus'], 3), 'concurrent here', round(st['concurrent_ms_per_step_here'], 2), 'with per-launch events', round(st['device_ms_mean_with_per_launch_events'], 3))
    if 'combine_ab' in s: print('  combine_ab', s['combine_ab'])
