"""DESIGN.md = tools/design/DESIGN.tpl.md with every @NAME@ replaced by a number read from profiles/r06/ (run from the repo root after the
evidence files have been refreshed: `python tools/design/fill_design.py`).  The prose lives in the template; edit it there."""
import json, re, sys
d='profiles/r06/'
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
 'FRONTEND': sp(fe['value']), 'FRONTEND_CPU': f(fe['cpu_baseline']['value'],1), 'K4ROC': f"average of {next(x['calls'] for x in json.load(open(d+'summary.json'))['kernel_trace_by_grid'] if 'vgicp_kernel' in x['kernel'])} launches {tr['kernel_avg_us_rocprof']:.1f} µs, its own bench line {ur['roofline']['kernel_ms']*1e3:.1f} µs, `profiles/r06/summary.json`",
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
# ---- round 6 additions ----
sl=b['single_factor_loop']; tl=sl['resident_timeline_us']; byv=sl['us_per_call_by_variant']; rc=sl['resident_session_cost']
natf=json.load(open(d+'bench_global256_native.json')); virt=natf['virtual_world8']; vb=virt['exchange_behind_the_call']; vs=virt['call_waits_for_the_exchange']
vb0=vb['host_breakdown_us']['device_0_caller_thread']; vbw=vb['host_breakdown_us']['worker_threads_max']
smr=json.load(open(d+'bench_submap20.json'))['roofline']
kr=rg['roofline']; kf=fe['roofline']
m1c=json.load(open(d+'m1_counters.json')); dv=m1c.get('derived',{}); ca=m1c['counters_avg_per_dispatch']
ul=json.load(open(d+'bench_odometry_under_load.json'))['config']['wirings']
hdr=open('include/glim_amd.h').read(); dhdr=open('include/glim_amd_diag.h').read()
n_stable=len(set(re.findall(r"\b(glim_amd_[a-z0-9_]+)\s*\(",hdr))); n_diag=len(set(re.findall(r"\b(glim_amd_[a-z0-9_]+)\s*\(",dhdr)))
def pct_or(x,nd=2): return 'n/a' if x is None else f(x,nd)
def ul_cell(w,mode):
    e=ul[w][mode]; return f"{e['idle']['p99_us']:.1f} → {e['under_load']['p99_us']:.0f} µs ({e['p99_ratio']:.1f}×)"
ul_rows=[('own context, high priority (shipped: the odometry module\'s)','own_context_high_priority','resident session: '),('own context, default priority','own_context','(no session) '),('one shared context (round 3)','shared_context','(no session) ')]
ul_table='| wiring | default switches | `resident=0` |\n|---|---|---|\n'+'\n'.join(f"| {name} | {pre}{ul_cell(w,'resident_session')} | {ul_cell(w,'launch_per_call')} |" for name,w,pre in ul_rows)
mt=ul['own_context_high_priority']['resident_session'].get('mapping_thread',{}); mtl=ul['own_context_high_priority']['launch_per_call'].get('mapping_thread',{})
bursts=one.get('frames_above_1p4x_median',{}); slowest=one.get('slowest_frames',[])
burst_txt=(f"{bursts.get('count',0)} of {bursts.get('of',0)} frames above 1.4 × the median, in {len(bursts.get('runs',[]))} run(s) "+', '.join(f"[frames {r[0]}–{r[1]} at {r[2]:.0f} ms]" for r in bursts.get('runs',[])[:6])+"; the slowest frames' excess sits in "+', '.join(sorted(set(x['excess_in'] for x in slowest)))) if bursts else 'n/a'
p99s=one.get('stage_p99_us',{})
transit=tl['host_round_trip']-tl['device_span']
def counters_txt(c):
    a=c['counters_avg_per_dispatch']; dvv=c.get('derived',{}); cyc=a['SQ_BUSY_CYCLES']/32.0; per_simd=a['SQ_INSTS_VALU']/1024.0
    return (f"{a['SQ_INSTS_VALU']/max(1,a['SQ_WAVES']):.0f} vector instructions per wavefront, {cyc/1e3:.0f} k cycles per launch at {cyc/c['kernel_avg_us_rocprof']/1e3:.2f} GHz, "
            f"one vector instruction issued per {cyc/per_simd:.2f} cycles of a SIMD, wavefronts waiting on a counter {100*(dvv.get('share_of_wave_cycles_waiting_on_any_counter') or 0):.0f} % of their resident "
            f"cycles, L2 hit rate {100*(dvv.get('l2_hit_rate') or 0):.0f} % of {a.get('TCC_REQ_sum',0)/1e6:.0f} M requests"), cyc/per_simd
m2c=json.load(open(d+'m2_counters.json')); m2c_txt,m2_cpv=counters_txt(m2c); m1c_txt,m1_cpv=counters_txt(m1c)
loc=json.load(open(d+'probe/m2_locality_probe.json'))
loc_rep=sum(v['ns_per_1000_visits'] for k,v in loc.items() if k.startswith('sample64_x'))/max(1,sum(1 for k in loc if k.startswith('sample64_x')))
loc_shuf=sum(v['ns_per_1000_visits'] for k,v in loc.items() if k.startswith('sample64_shuffled'))/max(1,sum(1 for k in loc if k.startswith('sample64_shuffled')))
vals.update({
 'CPU_THREADS': str(cb['cores']), 'M2_PTS': sp(m2['config']['mean_points_per_submap']),
 'SYNC_RES': f(byv['resident_session'],1), 'SYNC_SINGLE': f(byv['default_context_no_session'],1), 'SYNC_TWO': f(byv['two_dispatches'],1),
 'SYNC_VERDICT': ('met on this box' if byv['resident_session']<=10.5 else f"not met on this box ({byv['resident_session']:.1f} µs; 10.76–10.84 on the box of the step-major A/B, whose round-5 form would have taken ≈ 13.7)"),
 'TL_PUB': f(tl['leader_published'],2), 'TL_POSE': f(tl['worker_pose_seen_median'],2), 'TL_ROWS': f(tl['worker_row_computed_max'],2), 'TL_PUBROW': f(tl['worker_row_published_max'],2),
 'TL_LOOP': f(tl.get('worker_loop_left_median',0),2), 'TL_LOOPMAX': f(tl.get('worker_loop_left_max',0),2), 'TL_GSUM': f(tl.get('finaliser_group_sums_added',0),2), 'TL_ROT': f(tl.get('finaliser_blocks_rotated',0),2),
 'TL_CLOCK': f(tl.get('shader_clock_mhz',0),0), 'SPEEDUP': f(b['speedup_vs_cpu_baseline'],1),
 'TL_SUM': f(tl['finaliser_rows_summed'],2), 'TL_REC': f(tl['finaliser_record_stored'],2), 'TL_HOST': f(tl['host_round_trip'],2), 'TL_TRANSIT': f(transit,1),
 'COST128': f(rc['batched_128_factor_kernel_ms']['slowdown'],2), 'COST8': f(rc['8_factor_kernel_ms']['slowdown'],2),
 'VIRT_MS': f(vb['ms_per_evaluation'],2), 'VIRT_WAIT_BEHIND': f(vb['host_wait_for_the_exchange_after_the_call_us'],0), 'VIRT_WAIT_SYNC': f(vs['host_wait_for_the_exchange_after_the_call_us'],0),
 'VIRT_EXCH': f"{min(x['exchange_after_the_last_kernel_ms'] for x in vb['per_device']):.1f}–{max(x['exchange_after_the_last_kernel_ms'] for x in vb['per_device']):.1f}",
 'VIRT_HOST': f"post {vb0['post']:.0f} µs, the workers start {vbw['wake']:.0f} µs after the call, pose staging ≤ {max(vb0['pose_stage'],vbw['pose_stage']):.0f}, enqueue ≤ {max(vb0['enqueue'],vbw['enqueue']):.0f}, barrier ≤ {max(vb0['barrier'],vbw['barrier']):.0f}, exchange enqueue ≤ {max(vb0['collective'],vbw['collective']):.0f} µs",
 'SUBMAP_FRAC': f(smr['frac'],2), 'SUBMAP_ALGO': f(smr['algorithmic_48B_ratio_to_peak'],2),
 'KNN_RGBD_US': f(kr['kernel_ms']*1e3,1), 'KNN_RGBD_ROC': pct_or(kr.get('kernel_avg_us_rocprof'),1), 'KNN_RGBD_TRAFFIC': pct_or(kr['traffic']/1e6 if kr.get('traffic') else None,1), 'KNN_RGBD_ALGO': f(kr['algorithmic_bytes_per_launch']/1e6,1), 'KNN_RGBD_FRAC': f(kr['frac'],3),
 'KNN_FE_N': sp(kf['points']), 'KNN_FE_US': f(kf['kernel_ms']*1e3,1), 'KNN_FE_TRAFFIC': pct_or(kf['traffic']/1e6 if kf.get('traffic') else None,2), 'KNN_FE_FRAC': f(kf['frac'],4),
 'FRONT_RETIRE': f(fe['config']['stage_ms']['retire_the_frame_before_last'],3), 'FRONT_LIN': f(fe['config']['stage_ms']['linearize'],3),
 'M1_COUNTERS': m1c_txt,
 'M2_COUNTERS': m2c_txt, 'M1_CPV': f(m1_cpv,2), 'M2_CPV': f(m2_cpv,2), 'M1_ISSUE_PCT': f(100*2.85/m1_cpv,0), 'M2_ISSUE_PCT': f(100*2.74/m2_cpv,0),
 'LOC_ALL': f(loc['all_pairs']['ns_per_1000_visits'],2), 'LOC_REP': f(loc_rep,2), 'LOC_SHUF': f(loc_shuf,2), 'LOC_ONE': f(loc['one_pair_x32640']['ns_per_1000_visits'],2),
 'LOC_GAIN': f(100*(1-loc_rep/loc['all_pairs']['ns_per_1000_visits']),1),
 'FRAME_BURSTS': burst_txt, 'FRAME_P99_STAGES': ', '.join(f"{k.replace('_',' ')} {v:.0f}" for k,v in p99s.items()) if p99s else 'n/a',
 'UL_TABLE': ul_table,
 'UL_MAPPING': (f"{mt.get('loops_per_s_alone',0):.0f} loops/s alone, {mt.get('loops_per_s_beside_the_odometry',0):.0f} beside the odometry's resident session ({mt.get('slowdown',0):.2f}×), {mtl.get('loops_per_s_beside_the_odometry',0):.0f} beside its launch-per-call form ({mtl.get('slowdown',0):.2f}×)") if mt else 'n/a',
 'ABI_STABLE': str(n_stable), 'ABI_DIAG': f"{n_diag} entry points",
})
s=open('tools/design/DESIGN.tpl.md').read()
missing=set(re.findall(r'@([A-Z0-9_]+)@',s))-set(vals)
assert not missing, missing
for k,v in vals.items(): s=s.replace('@'+k+'@',v)
if '--check' in sys.argv:  # tests/test_abi_cpu.py: DESIGN.md is exactly the template filled from the committed evidence
    sys.exit(0 if open('DESIGN.md').read()==s else 1)
open('DESIGN.md','w').write(s)
for k in sorted(vals): print(k,'=',vals[k])
