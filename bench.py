#!/usr/bin/env python3
"""Headline benchmark: FQL gradient-steps/s at per-GPU batch 256 (BASELINE.json metric).

Workload = BASELINE.json configs[1]: antmaze-large-shaped synthetic offline data (obs=29, act=8),
batch=256, hidden 512x4, flow_steps=10, alpha=10, 1,000,000 device-resident transitions
(generator: SURVEY.md 8d), indices and the five noise tensors drawn by the engine's device RNG.
One "step" = one full FQLAgent.update (forward, backward, grad stats, Adam, Polyak) on one batch.

N > 1 (launched by torch.distributed.run): one process per GPU, replay sharded PHYSICALLY by transition index
(each rank uploads only its rows), gradient all-reduce over RCCL, identical optimizer step on every rank
("scaling": "weak").

Prints ONE JSON line on rank 0.  Besides the contract fields:
  roofline      per-kernel: the kernel family with the largest device time per update, its algorithmic FLOPs per launch
                (GEMM-shaped tasks, SURVEY.md 8d accounting) over its average launch duration, measured HERE with HIP events on
                the engine's stream (fql_profile_update: the update's launches issued in program order on one stream, a start / stop
                event pair attached to each dispatch - the dispatch duration rocprofv3 --kernel-trace reports, same serialised view;
                profiles/r02_kernel_stats.csv must agree);
                `traffic` / `mfma_util` come from the committed rocprofv3 --pmc summary (profiles/r02_pmc_summary.json), collected
                in separate passes as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 for 16-byte-per-lane reads on gfx950)
  whole_update  the same accounting over the whole update on the fenced wall clock
  kernels       every kernel family of the update: launches, average us, share of the serialised device time
  cpu_baseline  the torch-CPU restatement of the reference update (oracle/, kind "port") on this host's cores, bounded sample
  precision_bf16x3  (N = 1, state workload, headline run in fp32) the same workload, steps and warm-up on a second agent created with
                precision='bf16x3' (fql_config.precision = 2: split-bf16 products on the bf16 matrix cores, fp32 accumulation): its rate,
                its dominant kernel against the bf16 peak / 3, and its loss delta against the fp64 oracle.  Never `value`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PMC_FILE, PMC_FILE_BF16X3 = 'r03_pmc_summary.json', 'r03_pmc_summary_bf16x3.json'   # committed rocprofv3 --pmc summaries (tools/profile.sh)
FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (matrix), v_mfma_f32_16x16x4_f32
BF16_MATRIX_PEAK_TFLOPS = 2516.6  # dense bf16 matrix peak; a bf16x3 product is three bf16 MFMAs: peak / 3 = 838.9 algorithmic TFLOP/s
OP_NAMES = ['fql_gemm16_kernel', 'fql_side_kernel', 'fql_wgrad_kernel', 'fql_lnbwd_kernel', 'fql_prep_kernel', 'fql_post_onestep_kernel',
            'fql_euler_finish_kernel', 'fql_euler_persistent_kernel', 'fql_loss_critic_kernel', 'fql_loss_q_kernel', 'fql_loss_bc_kernel',
            'fql_loss_actor_kernel', 'fql_conv_wprep_kernel', 'fql_conv3x3_kernel', 'fql_conv3x3_u8_kernel', 'fql_maxpool_kernel',
            'fql_maxpool_bwd_kernel', 'fql_conv_wgrad_kernel', 'fql_conv_wgrad_reduce_kernel', 'fql_enc_dz_kernel', 'fql_chain_kernel',
            'fql_wfrag_kernel', 'fql_xchain_kernel', 'fql_head_dgrad_kernel', 'fql_dgrad0_kernel', 'fql_adam_kernel', 'fql_finalize_kernel']


def cpu_baseline(cfg, od, ad, B, budget_s=18.0, img=None):
    """Times the torch-CPU restatement of the reference update (oracle/, "port") on this host.  img = (H, W, C): visual agent."""
    import torch
    from oracle import fql_oracle as O
    from oracle.fql_oracle_torch import TorchFQL
    cores = torch.get_num_threads()
    params = O.init_params(0, img if img else od, ad, dict(cfg))
    ref = TorchFQL(params, dict(cfg), torch.float32)
    rng = np.random.default_rng(1)
    if img:
        def vb():
            return {'observations': rng.integers(0, 256, size=(B,) + tuple(img), dtype=np.uint8),
                    'next_observations': rng.integers(0, 256, size=(B,) + tuple(img), dtype=np.uint8),
                    'actions': rng.uniform(-1, 1, size=(B, ad)).astype(np.float32),
                    'rewards': -np.ones(B, np.float32), 'masks': np.ones(B, np.float32)}
        batches = [(vb(), O.make_noise(B, ad, 10 + i)) for i in range(2)] * 2
    else:
        ds = O.make_synthetic_dataset(8192, od, ad, seed=0)
        batches = [(O.sample_batch(ds, rng.integers(0, 8192, size=B)), O.make_noise(B, ad, 10 + i)) for i in range(4)]

    def run(budget, cap):
        ref.update(*batches[0])  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            ref.update(*batches[n % 4])
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget or n >= cap:
                return n, dt
    # thread counts: 1, 16, and every core torch sees; the fastest is the reported baseline (with the matrices of
    # this path - 256 x 512 x 512 - a 128-thread pool is slower than one thread)
    scan = {}
    for th, budget, cap in ((1, 5.0, 40), (min(16, cores), 5.0, 60), (cores, budget_s - 10.0, 200)):
        if th in scan:
            continue
        torch.set_num_threads(th)
        n, dt = run(budget, cap)
        scan[th] = (n / dt, n, dt)
    torch.set_num_threads(cores)
    best = max(scan, key=lambda t: scan[t][0])
    rate, n, dt = scan[best]
    return {'value': round(rate, 3), 'unit': 'grad-steps/s', 'cores': int(best), 'kind': 'port',
            'sample': f'{n} updates of the torch-CPU restatement (oracle/fql_oracle_torch.py, fp32, B={B}) in {dt:.1f}s on {best} thread(s); '
                      'CPU restatement of the reference path, not JAX',
            'threads_scan': {str(t): round(v[0], 3) for t, v in scan.items()}, 'host_cores': int(cores)}


def profile_kernels(agent, B, reps, split=False):
    """Per-launch device times (HIP events on the engine's stream) of `reps` updates -> per kernel family statistics."""
    names = list(OP_NAMES)
    if split:   # precision = 2 launches the split bodies for the side lanes and the Euler chain (the tail's input-gradient chain launches stay fp32)
        names[OP_NAMES.index('fql_side_kernel')] = 'fql_side_split_kernel'
        names[OP_NAMES.index('fql_chain_kernel')] = 'fql_chain_split_kernel (+ 3 fp32 fql_chain_kernel dgrad launches)'
        names[OP_NAMES.index('fql_conv3x3_kernel')] = 'fql_conv3x3_split_kernel'
        names[OP_NAMES.index('fql_conv_wgrad_kernel')] = 'fql_conv_wgrad_split_kernel'
        names[OP_NAMES.index('fql_conv3x3_u8_kernel')] = 'fql_conv3x3_u8_split_kernel'
    from fql_amd import _cabi
    lib = _cabi.load()
    f = lib.fql_profile_update
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6
    cap = 1024
    typ, lane, grid = (C.c_int * cap)(), (C.c_int * cap)(), (C.c_int * cap)()
    us, macs, null_us = (C.c_float * cap)(), (C.c_double * cap)(), C.c_float()
    fam, nulls = {}, []
    for r in range(reps + 2):
        n = f(agent._h, B, cap, typ, lane, grid, us, macs, C.byref(null_us))
        if n <= 0:
            return None
        if r < 2:
            continue   # warm-up passes
        nulls.append(null_us.value)
        # the start / stop events ride on each dispatch (hipExtLaunchKernelGGL): their elapsed time is the dispatch's own duration, the
        # quantity `rocprofv3 --kernel-trace --stats` averages - nothing to calibrate
        gap = 0.0
        for i in range(n):
            d = fam.setdefault(names[typ[i]], {'launches': 0, 'us': 0.0, 'macs': 0.0, 'min_us': 1e9, 'max_us': 0.0, 'raw': 0.0})
            t = max(0.5, us[i] - gap)
            d['launches'] += 1; d['us'] += t; d['macs'] += macs[i]; d['raw'] += us[i]
            d['min_us'] = min(d['min_us'], t); d['max_us'] = max(d['max_us'], t)
    tot = sum(d['us'] for d in fam.values())
    out = {}
    for k, d in fam.items():
        out[k] = {'launches_per_update': d['launches'] / reps, 'avg_us': d['us'] / d['launches'], 'us_per_update': d['us'] / reps,
                  'share': d['us'] / tot, 'flop_per_launch': 2.0 * d['macs'] / d['launches'], 'min_us': d['min_us'], 'max_us': d['max_us'],
                  'event_interval_us': d['raw'] / d['launches'], 'null_interval_us': float(np.median(nulls))}
    return out


def pmc_counters(fname, kernel):
    """Static rocprofv3 --pmc numbers of `kernel` from a committed summary (collected in separate passes, tools/profile.sh): returned with the
    hash of the kernel sources they were measured on and `traffic_stale` = the sources here differ from those."""
    path = os.path.join(ROOT, 'profiles', fname)
    if not os.path.exists(path):
        return {}
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        from pmc_summary import source_sha16
        pj = json.load(open(path))
        k = pj.get('kernels', {}).get(kernel)
        if not k:
            return {}
        return {'traffic': k.get('hbm_bytes_per_launch'), 'mfma_util': k.get('mfma_util'), 'traffic_source': pj.get('source'),
                'traffic_file': 'profiles/' + fname, 'traffic_source_sha16': pj.get('source_sha16'),
                'traffic_stale': pj.get('source_sha16') != source_sha16(ROOT)}
    except Exception:
        return {}


def short_window(agent, B, steps, warmup, torch):
    for _ in range(warmup):
        agent.update_from_dataset(B)
    agent.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        agent.update_from_dataset(B)
    agent.synchronize(); torch.cuda.synchronize()
    return time.perf_counter() - t0


def other_configs(torch, peak_tf):
    """The other single-GPU configurations of BASELINE.json in the same process, short fenced windows: configs[2] (obs 40, act 4, B 1024, alpha 300) and
    configs[4] (visual: impala_small, 64x64x9, B 256, alpha 300).  fp32.  Each: rate, ms per update, its dominant kernel against the fp32 matrix peak."""
    import fql_amd
    from fql_amd.synthetic import make_synthetic_dataset, make_synthetic_frames
    res = []
    for name in ('configs[2]', 'configs[4]'):
        cfg = fql_amd.get_config()
        if name == 'configs[2]':
            od, ad, B, steps, warm = 40, 4, 1024, 200, 20
            cfg.update(alpha=300.0, batch_size=B)
            ds = make_synthetic_dataset(200_000, od, ad, seed=0)
            agent = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], cfg)
            agent.upload_dataset(ds)
            work = 'synthetic replay (obs=40, act=4), batch=1024, hidden=512x4, flow_steps=10, alpha=300, 200k device-resident transitions (BASELINE.json configs[2])'
        else:
            ad, B, steps, warm = 5, 256, 60, 10
            cfg.update(alpha=300.0, batch_size=B, encoder='impala_small')
            ds = make_synthetic_frames(4000, ad, seed=0)
            agent = fql_amd.FQLAgent.create(0, np.zeros((1, 64, 64, 9), np.uint8), ds['actions'][:1], cfg)
            agent.upload_dataset(ds, frame_stack=3, p_aug=0.5)
            work = 'visual-cube-shaped synthetic replay (uint8 64x64x9, act=5), impala_small encoders, batch=256, alpha=300, p_aug=0.5, 4000 device-resident frames (BASELINE.json configs[4])'
        dt = short_window(agent, B, steps, warm, torch)
        st = agent.stats()
        r = {'workload': work, 'value': round(steps / dt, 2), 'unit': 'grad-steps/s', 'ms_per_step': round(dt * 1e3 / steps, 4), 'steps': steps, 'warmup': warm,
             'dtype': 'f32', 'samples_per_s': round(steps / dt * B, 1), 'kernel_launches_per_update': st['launches_per_update'],
             'whole_update_tflops': round(2.0 * st['macs_per_update'] * steps / dt / 1e12, 3)}
        fams = profile_kernels(agent, B, reps=5)
        if fams:
            mm = {k: v for k, v in fams.items() if v['flop_per_launch'] > 0}
            dom = max(mm, key=lambda k: mm[k]['us_per_update'])
            ach = mm[dom]['flop_per_launch'] / (mm[dom]['avg_us'] * 1e-6) / 1e12
            r['roofline'] = {'kernel': dom, 'achieved': round(ach, 3), 'peak': peak_tf, 'unit': 'TFLOP/s', 'frac': round(ach / peak_tf, 4),
                             'avg_launch_us': round(mm[dom]['avg_us'], 3), 'launches_per_update': round(mm[dom]['launches_per_update'], 2)}
        agent.close()
        res.append(r)
    return res


def bench_bf16x3(args, cfg, ds, od, ad, B, torch):
    """Second agent, precision='bf16x3', same workload / steps / warm-up / fences as the headline; never the headline `value`."""
    import fql_amd
    c2 = fql_amd.get_config()
    c2.update({k: v for k, v in dict(cfg).items() if k not in ('ob_dims', 'action_dim')})
    c2['precision'] = 'bf16x3'
    agent = fql_amd.FQLAgent.create(0, ds['observations'][:1], ds['actions'][:1], c2)
    agent.upload_dataset(ds)
    warm = args.warmup   # (the headline's own warm-up: the two rates in one line are like for like)
    for _ in range(warm):
        agent.update_from_dataset(B)
    torch.cuda.synchronize(); agent.read_info()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        agent.update_from_dataset(B)
    agent.read_info(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = agent.stats()
    rate = args.steps / dt
    flop = 2.0 * st['macs_per_update']
    whole = flop * rate / 1e12
    peak = BF16_MATRIX_PEAK_TFLOPS / 3.0
    res = {'value': round(rate, 2), 'unit': 'grad-steps/s', 'ms_per_step': round(dt * 1e3 / args.steps, 5), 'steps': args.steps, 'warmup': warm,
           'dtype': 'bf16x3 (fp32 operands split into hi + lo bf16, a b = a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate; '
                    'parameters, activations, gradients, Adam in fp32)',
           'whole_update': {'achieved_tflops': round(whole, 3), 'frac_of_bf16_matrix_peak_over_3': round(whole / peak, 4),
                            'kernel_launches_per_update': st['launches_per_update']}}
    fams = profile_kernels(agent, B, reps=20, split=True)
    if fams:
        mm = {k: v for k, v in fams.items() if v['flop_per_launch'] > 0}
        dom = max(mm, key=lambda k: mm[k]['us_per_update'])
        d = mm[dom]
        ach = d['flop_per_launch'] / (d['avg_us'] * 1e-6) / 1e12
        res['roofline'] = {'bound': 'mfma', 'achieved': round(ach, 3), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                           'traffic': None, 'kernel': dom, 'flop_per_launch': round(d['flop_per_launch']), 'avg_launch_us': round(d['avg_us'], 3),
                           'launches_per_update': round(d['launches_per_update'], 2),
                           'peak_note': 'dense bf16 matrix peak 2516.6 TFLOP/s / 3 MFMAs per product; the kernel is latency-bound, not MFMA-bound'}
        res['roofline'].update(pmc_counters(PMC_FILE_BF16X3, dom))
        if res['roofline'].get('traffic'):
            res['roofline']['hbm_gbps'] = round(res['roofline']['traffic'] / (d['avg_us'] * 1e-6) / 1e9, 1)
        res['kernels'] = {k: {'launches': round(v['launches_per_update'], 2), 'avg_us': round(v['avg_us'], 2), 'share': round(v['share'], 4)}
                          for k, v in sorted(fams.items(), key=lambda kv: -kv[1]['us_per_update'])}
    if not args.no_cpu_baseline:
        from oracle import fql_oracle as O   # checker only
        pb = O.sample_batch(ds, np.random.default_rng(3).integers(0, args.rows, size=B))
        pn = O.make_noise(B, ad, 4)
        ref = O.OracleFQL(agent.get_params(), {k: v for k, v in dict(c2).items() if k != 'rng'}, od, ad, np.float64)
        lg, ig = agent.total_loss(pb, None, noise=pn)
        lr, ir = ref.total_loss(pb, pn)
        res['loss_delta'] = {'total_loss_gpu': round(float(lg), 6), 'total_loss_oracle': round(float(lr), 6), 'abs_delta': float(abs(lg - lr)),
                             'max_abs_delta_info': float(max(abs(ig[k] - ir[k]) for k in ir)),
                             'per_info': {k: float(ig[k] - ir[k]) for k in ir}}
    agent.close()
    return res


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks of this script (one per GPU, RCCL over xGMI) with the
    torch.distributed environment the launcher would have set, relay rank 0's ONE json line on stdout, return the worst exit code.
    The parent touches no GPU (a process that has initialised HIP must not exec or fork GPU children on this pool)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3000)
    ap.add_argument('--warmup', type=int, default=300)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--rows', type=int, default=1_000_000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the per-kernel event pass and the host-batch rate (profiling runs)')
    ap.add_argument('--workload', choices=['state', 'visual'], default='state',
                    help="state = BASELINE.json configs[1] (the headline metric); visual = configs[4] (impala_small, 64x64x9 uint8)")
    ap.add_argument('--frames', type=int, default=20_000, help='frames in the synthetic visual dataset')
    ap.add_argument('--obs-dim', type=int, default=29, help='state workload: observation width (configs[2] uses 40)')
    ap.add_argument('--act-dim', type=int, default=8, help='state workload: action width (configs[2] uses 4)')
    ap.add_argument('--alpha', type=float, default=None, help='BC coefficient (default 10 for the state workload, 300 visual)')
    ap.add_argument('--precision', choices=['fp32', 'bf16x3'], default=os.environ.get('FQL_BENCH_PRECISION', 'fp32'),
                    help="fp32 = fp32 matrix cores (the headline); bf16x3 = split-bf16 products on the bf16 matrix cores, fp32 accumulate "
                         "(fql_config.precision = 2; reported against the bf16 peak / 3)")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))   # (this process has made no GPU call: it only starts the ranks and relays rank 0's line)
    if os.environ.get('FQL_BENCH_SPAWN_DRYRUN'):    # tests of the launcher path (no GPU): a rank reports the environment it was started with
        if os.environ.get('RANK', '0') == '0':
            print(json.dumps({k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')} | {'gpus': args.gpus}))
        return

    import torch
    import fql_amd
    from fql_amd.parallel import DataParallelFQL
    from fql_amd.synthetic import make_synthetic_dataset, make_synthetic_frames

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks for --gpus N (or none: bench.py starts them itself)')
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f'rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) visible')
    torch.cuda.set_device(local_rank)
    dist = None
    force_dp = bool(os.environ.get('FQL_BENCH_FORCE_DP'))  # exercise the RCCL path on one GPU (torchrun --nproc-per-node 1)
    # RCCL prints a version banner on STDOUT at communicator creation; the contract is ONE json line there, so file descriptor 1 points
    # at stderr until the line is printed
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if world > 1 or force_dp:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    od, ad, B = args.obs_dim, args.act_dim, args.batch
    peak_tf = FP32_MATRIX_PEAK_TFLOPS if args.precision == 'fp32' else BF16_MATRIX_PEAK_TFLOPS / 3.0
    cfg = fql_amd.get_config()
    visual = args.workload == 'visual'
    seed = 0    # the same seed on every rank: DataParallelFQL mixes the rank into the device RNG stream
    if visual:
        # BASELINE.json configs[4] / SURVEY.md 8d "Config 5": uint8 frames, frame_stack 3 -> [64, 64, 9], impala_small encoders,
        # alpha 300, p_aug 0.5; act_dim is a runtime parameter (cube-single: 5)
        ad = 5
        args.rows = args.frames
        ds = make_synthetic_frames(args.frames, ad, seed=0)
        cfg.update(alpha=300.0 if args.alpha is None else args.alpha, batch_size=B, encoder='impala_small', precision=args.precision)
        agent = fql_amd.FQLAgent.create(seed, np.zeros((1, 64, 64, 9), np.uint8), ds['actions'][:1], cfg)
        up_kw = dict(frame_stack=3, p_aug=0.5)
    else:
        cfg.update(alpha=10.0 if args.alpha is None else args.alpha, batch_size=B, precision=args.precision)
        ds = make_synthetic_dataset(args.rows, od, ad, seed=0)
        agent = fql_amd.FQLAgent.create(seed, ds['observations'][:1], ds['actions'][:1], cfg)
        up_kw = {}
    dp = DataParallelFQL(agent, overlap=os.environ.get('FQL_DP_OVERLAP', '1') != '0') if dist is not None else None
    dp_mode = None
    if dp is not None:
        dp.upload_shard(ds, **up_kw)          # this rank's rows only
        # nobody is there to choose between the overlapped (bucketed, two streams) and the plain (one all-reduce) step at this world size:
        # time both with the collectives live and keep the faster (FQL_DP_OVERLAP=0 / 1 pins it)
        dp_mode = dp.autotune(batch_size=B, steps=40) if 'FQL_DP_OVERLAP' not in os.environ else {'chosen': 'overlapped' if dp.buckets is not None else 'plain', 'pinned': True}
    else:
        agent.upload_dataset(ds, **up_kw)

    def step():
        if dp is not None:
            dp.update_from_dataset(batch_size=B)
        else:
            agent.update_from_dataset(B)      # the engine's own stream; no host synchronisation inside the window

    dispatch = []

    def fence():
        dispatch.append(agent.synchronize())  # the engine's own hardware queues (stream-less updates) are not torch's: wait for them first
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    info = agent.read_info()
    st = agent.stats()

    if rank == 0:
        steps_per_s = args.steps / dt
        flop_per_step = 2.0 * st['macs_per_update']            # algorithmic: SURVEY.md 8d (12.376 GFLOP at B=256)
        step_us_wall = dt * 1e6 / args.steps                   # fenced wall clock per update
        whole = flop_per_step / (step_us_wall * 1e-6) / 1e12
        work = ('visual-cube-shaped synthetic replay (uint8 64x64x9 = 3 stacked frames, act=5), impala_small encoders, '
                f'batch={B}/GPU, hidden=512x4, flow_steps=10, alpha={cfg["alpha"]:g}, p_aug=0.5, {args.frames} device-resident frames '
                '(BASELINE.json configs[4])') if visual else (
            f'antmaze-large-shaped synthetic replay (obs={od}, act={ad}), batch={B}/GPU, hidden=512x4, '
            f'flow_steps=10, alpha={cfg["alpha"]:g}, 1M device-resident transitions (BASELINE.json configs[1])'
            if (od, ad, B) == (29, 8, 256) else
            f'synthetic replay (obs={od}, act={ad}), batch={B}/GPU, hidden=512x4, flow_steps=10, '
            f'alpha={cfg["alpha"]:g}, device-resident transitions (BASELINE.json configs[2] shape when 40/4/1024)')
        out = {
            'metric': 'FQL gradient-steps/s (batch=256)', 'value': round(steps_per_s * world, 2), 'unit': 'grad-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt * 1e3 / args.steps, 5),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32' if args.precision == 'fp32' else 'bf16x3', 'data': 'synthetic',
            'config': {'workload': work, 'parallelism': f'dp{world}', 'samples_per_s': round(steps_per_s * world * B, 1)},
            'dispatch': ('AQL packets on the engine\'s own HSA queues (fql_amd/csrc/fql_aql.h)' if dispatch[-1] == 'aql' else 'captured hipGraph on HIP streams'),
            'rccl_ranks': (dist.get_world_size() if dist is not None else 1), 'backend': (dist.get_backend() if dist is not None else None),
            'device': torch.cuda.current_device(), 'data_parallel_step': dp_mode,
            'whole_update': {'flop': flop_per_step, 'wall_us': round(step_us_wall, 3), 'achieved_tflops': round(whole, 3),
                             ('frac_of_fp32_matrix_peak' if peak_tf == FP32_MATRIX_PEAK_TFLOPS else 'frac_of_bf16_matrix_peak_over_3'): round(whole / peak_tf, 4),
                             'kernel_launches_per_update': st['launches_per_update']},
            'last_info': {k: round(v, 5) for k, v in info.items()},
        }
        roof = {'bound': 'mfma', 'achieved': round(whole, 3), 'peak': peak_tf, 'unit': 'TFLOP/s',
                'frac': round(whole / peak_tf, 4), 'traffic': None, 'kernel': 'whole update (per-kernel pass skipped)'}
        if world == 1 and not args.no_extras:
            fams = profile_kernels(agent, B, reps=20, split=(args.precision == 'bf16x3'))
            if fams:
                mm = {k: v for k, v in fams.items() if v['flop_per_launch'] > 0}
                dom = max(mm, key=lambda k: mm[k]['us_per_update'])
                d = mm[dom]
                ach = d['flop_per_launch'] / (d['avg_us'] * 1e-6) / 1e12
                roof = {'bound': 'mfma', 'achieved': round(ach, 3), 'peak': peak_tf, 'unit': 'TFLOP/s',
                        'frac': round(ach / peak_tf, 4), 'traffic': None, 'kernel': dom,
                        'flop_per_launch': round(d['flop_per_launch']), 'avg_launch_us': round(d['avg_us'], 3),
                        'launches_per_update': round(d['launches_per_update'], 2),
                        'measured': 'start / stop HIP events attached to every dispatch (hipExtLaunchKernelGGL) of 20 updates issued in program '
                                    'order on the engine stream (fql_profile_update; serialised like rocprofv3 --kernel-trace): avg_launch_us = mean '
                                    'elapsed time of the pair = the dispatch duration a kernel trace reports'}
                if not visual:
                    roof.update(pmc_counters(PMC_FILE if args.precision == 'fp32' else PMC_FILE_BF16X3, dom))
                    if roof.get('traffic'):   # HBM-side bytes per launch (rocprofv3 --pmc, separate passes) over the launch duration measured here
                        roof['hbm_gbps'] = round(roof['traffic'] / (d['avg_us'] * 1e-6) / 1e9, 1)
                out['kernels'] = {k: {'launches': round(v['launches_per_update'], 2), 'avg_us': round(v['avg_us'], 2),
                                      'share': round(v['share'], 4),
                                      'tflops': round(v['flop_per_launch'] / (v['avg_us'] * 1e-6) / 1e12, 2) if v['flop_per_launch'] else None}
                                  for k, v in sorted(fams.items(), key=lambda kv: -kv[1]['us_per_update'])}
            if not visual:
                # the drop-in loop of main.py:201,216 with HOST batches (numpy -> staged by the engine, lazy infos): PCIe-inclusive rate
                idx = np.random.default_rng(1).integers(0, args.rows, size=(64, B))
                hb = [{k: v[i] for k, v in ds.items() if k != 'terminals'} for i in idx]
                for i in range(50):
                    agent.update(hb[i % 64])
                torch.cuda.synchronize(); agent.read_info()
                t1 = time.perf_counter()
                nh = 600
                for i in range(nh):
                    _, lazy = agent.update(hb[i % 64])
                agent.read_info()
                out['host_batch_update'] = {'value': round(nh / (time.perf_counter() - t1), 1), 'unit': 'grad-steps/s',
                                            'note': 'agent.update(numpy batch) with lazy info, ~70 KB staged H2D per call (PCIe-inclusive; never `value`)'}
        out['roofline'] = roof
        if world == 1 and not args.no_extras and not visual and args.precision == 'fp32' and not force_dp:
            out['precision_bf16x3'] = bench_bf16x3(args, cfg, ds, od, ad, B, torch)
            if (od, ad, B) == (29, 8, 256):
                out['other_configs'] = other_configs(torch, FP32_MATRIX_PEAK_TFLOPS)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, od, ad, B, img=(64, 64, 9) if visual else None)
            if not visual:
                # "loss delta vs the reference" (BASELINE.json metric), against the CPU restatement: one total_loss on the
                # engine's CURRENT parameters and a fixed synthetic batch with explicit noise, fp64 oracle (checker only)
                from oracle import fql_oracle as O
                pb = O.sample_batch(ds, np.random.default_rng(3).integers(0, args.rows, size=B))
                pn = O.make_noise(B, ad, 4)
                ref = O.OracleFQL(agent.get_params(), {k: v for k, v in dict(cfg).items() if k != 'rng'}, od, ad, np.float64)
                lg, ig = agent.total_loss(pb, None, noise=pn)
                lr, ir = ref.total_loss(pb, pn)
                out['cpu_baseline']['loss_delta'] = {'total_loss_gpu': round(float(lg), 6), 'total_loss_oracle': round(float(lr), 6),
                                                     'abs_delta': float(abs(lg - lr)),
                                                     'max_abs_delta_info': float(max(abs(ig[k] - ir[k]) for k in ir))}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
