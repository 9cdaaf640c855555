"""DESIGN.md = tools/design/DESIGN.tpl.md with every @NAME@ replaced by a number read from profiles/r05/ (run from the repo root after the
evidence files have been refreshed: `python tools/design/fill_design.py`).  The prose lives in the template; edit it there."""
import json, re, sys
d='profiles/r05/'
b=json.load(open(d+'bench.json')); m2=b['m2_global256']; n=m2['native_c_abi_world1']; r=b['roofline']; rf=m2['roofline']
o=json.load(open(d+'bench_odometry_frame.json'))['config']['frames_10000_pts']
ll=o['live_loop']; one=ll['one_submission_create_frame']; sep=ll['separate_calls']
u=json.load(open(d+'bench_odometry_under_load.json'))['config']['wirings']['own_context_high_priority']['resident_session']
sm=json.load(open(d+'bench_submap20.json'))['config']
rg=json.load(open(d+'bench_rgbd300k.json')); fe=json.load(open(d+'bench_frontend128k.json'))
ur=json.load(open(d+'bench_under_rocprof.json'))
tr=json.load(open(d+'traffic.json'))
vm=open(d+'voxelmap_time.txt').read()
vmv=[float(x) for x in re.findall(r'insert p50 ([0-9.]+) us',vm)]
ps=m2['predicted_scaling']; src=ps['pair_order_source_major']['cost_model_points']; tgt=ps['pair_order_target_major']['cost_model_points']
cb=b['cpu_baseline']; curve=cb['thread_curve_calls_per_s']
bd=json.load(open(d+'bench_global256_native.json'))['native']['host_breakdown_us']['device_0_caller_thread']
st=one['stage_p50_us']
sw=n['pieces_sweep']
f=lambda x,nd=2: f"{x:.{nd}f}"
sp=lambda x: f"{int(round(x)):,}".replace(',',' ')
nat_alone=json.load(open(d+'bench_global256_native.json'))['native']
cop=n.get('with_device_to_host_copies_behind_the_pieces',{}); onerank=n.get('with_the_one_rank_library_call_every_evaluation',{})
take2=ll['one_submission_as_round5_take2']; nofused=ll['one_submission_without_fused_frame_kernels']; nogate=ll['one_submission_without_gated_pull']; norec=ll['one_submission_without_plan_recycling']
fc=one['inside_frame_create_p50_us_since_entry']
vals={
 'NATIVE_COPIES': f(cop['ms_per_evaluation'],2), 'NATIVE_COPIES_DELTA': f"{cop['ms_per_evaluation']-n['ms_per_evaluation']:.2f}", 'NATIVE_COPIES_AFTER': f(cop['collective_and_copy_out_after_the_kernels_ms'][0]*1e3,0),
 'NATIVE_AFTER': f(n['per_device'][0]['collective_and_copy_out_after_the_kernels_ms']*1e3,0), 'NATIVE_KERNELS': f(n['per_device'][0]['kernels_ms'],2),
 'NATIVE_MINUS_M2': f"{n['ms_per_evaluation']-m2['ms_per_step']:+.2f}", 'ONE_RANK_HOST': f(onerank['library_calls_host_us'],1),
 'FRAME_NOFUSED_NOGATE': f(ll['one_submission_without_fused_frame_kernels_and_gated_pull']['frame_us']['p50'],0),
 'FRAME_TAKE2': f(take2['frame_us']['p50'],0), 'FRAME_NOFUSED': f(nofused['frame_us']['p50'],0), 'FRAME_NOGATE': f(nogate['frame_us']['p50'],0), 'FRAME_NORECYCLE': f(norec['frame_us']['p50'],0),
 'LIN_FIRST': f(st['first_linearisation_new_factor_list'],0), 'LIN_FIRST_NORECYCLE': f(norec['stage_p50_us']['first_linearisation_new_factor_list'],0),
 'PLANS_BUILT': str(one['factor_plans_built']), 'PLANS_RECYCLED': str(one['of_them_in_the_buffers_of_an_evicted_plan']),
 'FC_LAUNCHED': f(fc['pull_kernel_launched'],1), 'FC_PACKED': f(fc['host_conversion_done'],1), 'FC_SEEN': f(fc['completion_word_seen'],1), 'FC_TAIL': f(fc['completion_word_seen']-fc['host_conversion_done'],0),
 'FC_TAIL_BEFORE': '29',
 'FC_PACK_US': f(fc['host_conversion_done']-fc['pull_kernel_launched'],0),
 'ACHIEVED': f(r['achieved']/1000,2), 'ALGO_RATIO': f(r['algorithmic_48B_ratio_to_peak'],2), 'BATCHED': f(b['batched_calls_per_s']/1e6,2),
 'BOUND2': f(src['2']['compute_only_speedup_bound'],2), 'BOUND4': f(src['4']['compute_only_speedup_bound'],2), 'BOUND8': f(src['8']['compute_only_speedup_bound'],2),
 'BOUNDT8': f(tgt['8']['compute_only_speedup_bound'],2), 'BSPEEDUP': sp(b['batched_speedup_vs_cpu_baseline']), 'CPU': sp(cb['value']),
 'CPUMS': f(1e3/cb['value'],2), 'CURVE': ' / '.join(sp(curve[k]) for k in sorted(curve,key=int)), 'FP32FRAC': f(rf['fp32']['frac_of_fp32_vector_peak'],2),
 'FP32TF': f(rf['fp32']['achieved_tflops'],1), 'FRAC629': f(r['frac_of_6.29TBs_copy_rate'],2), 'FRAC': f(r['frac'],2), 'FRAC_M2': f(rf['frac'],2),
 'FRAME_ONE99': f(one['frame_us']['p99'],0), 'FRAME_ONE': f(one['frame_us']['p50'],0), 'FRAME_SEP99': f(sep['frame_us']['p99'],0), 'FRAME_SEP': f(sep['frame_us']['p50'],0),
 'FRAME_STAGES': f"clone + two maps {st['clone_and_two_voxelmaps']:.0f} µs, first linearisation of the new list {st['first_linearisation_new_factor_list']:.0f}, each further one {st['each_further_linearisation']:.0f}, overlap {st['overlap_15_targets']:.0f}, retiring a frame {st['retire_oldest_window_frame']:.0f}",
 'FRONTEND': sp(fe['value']), 'FRONTEND_CPU': f(fe['cpu_baseline']['value'],1), 'K4ROC': f"average of {next(x['calls'] for x in json.load(open(d+'summary.json'))['kernel_trace_by_grid'] if 'vgicp_kernel' in x['kernel'])} launches {tr['kernel_avg_us_rocprof']:.1f} µs, its own bench line {ur['roofline']['kernel_ms']*1e3:.1f} µs, `profiles/r05/summary.json`",
 'K4US': f(r['kernel_ms']*1e3,1), 'M2': f(m2['ms_per_step'],2), 'M2K': f(rf['kernel_ms'],2), 'NATIVE': f(n['ms_per_evaluation'],2),
 'NATIVE_BD': f"pose staging {bd['pose_stage']:.0f} µs + enqueue {bd['enqueue']:.0f} + collective enqueue {bd['collective']:.0f} (all beside running kernels except the first piece's share), {bd['wait']/1e3:.2f} ms waiting for the device, {bd['scan']:.1f} µs for the cost: {bd['total']/1e3:.2f} ms in the call",
 'NATIVE_ALONE': f(json.load(open(d+'bench_global256_native.json'))['native']['ms_per_evaluation'],2), 'NATIVE_ALONE_DELTA': f"{json.load(open(d+'bench_global256_native.json'))['native']['ms_per_evaluation']-m2['synchronous_per_evaluation']['ms_per_evaluation']:+.2f}",
 'NATIVE_DELTA': f"{m2['synchronous_per_evaluation']['native_minus_this_ms']:+.2f}", 'OVBATCH': f(o['keyframe_elimination_loop_one_batch_us'],0),
 'PIECES': ' / '.join(f(sw[k]['ms_per_evaluation'],2) for k in ('1','2') if k in sw)+' / '+f(sw['default']['ms_per_evaluation'],2)+' / '+f(sw['8']['ms_per_evaluation'],2),
 'RGBD': sp(rg['value']), 'SHARDSUM': f(src['8']['shards_sum_ms'],2), 'SPEEDUP': f(b['speedup_vs_cpu_baseline'],1),
 'SUBMAP': f"linearise {sm['bundle_linearize_ms']:.3f} ms per step, LM iteration {sm['lm_iteration_ms']:.2f} ms (round 4: 0.24–0.25 / 0.48)",
 'SYNC': f(m2['synchronous_per_evaluation']['ms_per_evaluation'],2), 'TRAFFIC_MB': f(r['traffic']/1e6,1),
 'UNDERLOAD': f"**{u['p99_ratio']:.1f}×** ({u['idle']['p99_us']:.0f} → {u['under_load']['p99_us']:.0f} µs)", 'VALUE': sp(b['value']),
 'VM10': f"{min(vmv[0],vmv[1]):.0f}–{max(vmv[0],vmv[1]):.0f}", 'VM131': f"{min(vmv[2],vmv[3]):.0f}–{max(vmv[2],vmv[3]):.0f}",
}
s=open('tools/design/DESIGN.tpl.md').read()
missing=set(re.findall(r'@([A-Z0-9_]+)@',s))-set(vals)
assert not missing, missing
for k,v in vals.items(): s=s.replace('@'+k+'@',v)
if '--check' in sys.argv:  # tests/test_abi_cpu.py: DESIGN.md is exactly the template filled from the committed evidence
    sys.exit(0 if open('DESIGN.md').read()==s else 1)
open('DESIGN.md','w').write(s)
for k in sorted(vals): print(k,'=',vals[k])
