#!/usr/bin/env python3
"""Headline benchmark: FQL gradient-steps/s at per-GPU batch 256 (BASELINE.json metric).

Workload = BASELINE.json configs[1]: antmaze-large-shaped synthetic offline data (obs=29, act=8),
batch=256, hidden 512x4, flow_steps=10, alpha=10, 1,000,000 device-resident transitions
(generator: SURVEY.md 8d), indices and the five noise tensors drawn by the engine's device RNG.
One "step" = one full FQLAgent.update (forward, backward, grad stats, Adam, Polyak) on one batch.

N > 1 (launched by torch.distributed.run): one process per GPU, replay sharded by transition index,
gradient all-reduce over RCCL, identical optimizer step on every rank ("scaling": "weak").

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (matrix), v_mfma_f32_16x16x4_f32


def cpu_baseline(cfg, od, ad, B, budget_s=20.0, img=None):
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
    ref.update(*batches[0])  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        ref.update(*batches[n % 4])
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 200:
            break
    return {'value': round(n / dt, 3), 'unit': 'grad-steps/s', 'cores': int(cores), 'kind': 'port',
            'sample': f'{n} updates of the torch-CPU restatement (oracle/fql_oracle_torch.py, fp32, B={B}) in {dt:.1f}s; '
                      'CPU restatement of the reference path, not JAX'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3000)
    ap.add_argument('--warmup', type=int, default=300)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--rows', type=int, default=1_000_000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', choices=['state', 'visual'], default='state',
                    help="state = BASELINE.json configs[1] (the headline metric); visual = configs[4] (impala_small, 64x64x9 uint8)")
    ap.add_argument('--frames', type=int, default=20_000, help='frames in the synthetic visual dataset')
    ap.add_argument('--obs-dim', type=int, default=29, help='state workload: observation width (configs[2] uses 40)')
    ap.add_argument('--act-dim', type=int, default=8, help='state workload: action width (configs[2] uses 4)')
    ap.add_argument('--alpha', type=float, default=None, help='BC coefficient (default 10 for the state workload, 300 visual)')
    args = ap.parse_args()

    import torch
    import fql_amd
    from fql_amd.parallel import DataParallelFQL, shard_range
    from oracle import fql_oracle as O  # synthetic data generator only

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...')
    torch.cuda.set_device(local_rank)
    dist = None
    force_dp = bool(os.environ.get('FQL_BENCH_FORCE_DP'))  # exercise the RCCL path on one GPU (torchrun --nproc-per-node 1)
    if world > 1 or force_dp:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    od, ad, B = args.obs_dim, args.act_dim, args.batch
    cfg = fql_amd.get_config()
    visual = args.workload == 'visual'
    if visual:
        # BASELINE.json configs[4] / SURVEY.md 8d "Config 5": uint8 frames, frame_stack 3 -> [64, 64, 9], impala_small encoders,
        # alpha 300, p_aug 0.5; act_dim is a runtime parameter (cube-single: 5)
        ad = 5
        n = args.rows = args.frames
        rng = np.random.default_rng(0)
        term = (rng.random(n) < 1.0 / 200).astype(np.float32); term[-1] = 1
        ds = {'observations': rng.integers(0, 256, size=(n, 64, 64, 3), dtype=np.uint8),
              'next_observations': rng.integers(0, 256, size=(n, 64, 64, 3), dtype=np.uint8),
              'actions': np.clip(rng.uniform(-1, 1, size=(n, ad)), -1 + 1e-5, 1 - 1e-5).astype(np.float32),
              'rewards': -(rng.random(n) < 0.99).astype(np.float32), 'masks': 1 - term, 'terminals': term}
        cfg.update(alpha=300.0, batch_size=B, encoder='impala_small')
        agent = fql_amd.FQLAgent.create(rank, np.zeros((1, 64, 64, 9), np.uint8), ds['actions'][:1], cfg)
        agent.upload_dataset(ds, frame_stack=3, p_aug=0.5)
    else:
        cfg.update(alpha=10.0 if args.alpha is None else args.alpha, batch_size=B)
        ds = O.make_synthetic_dataset(args.rows, od, ad, seed=0)
        agent = fql_amd.FQLAgent.create(rank, ds['observations'][:1], ds['actions'][:1], cfg)
        agent.upload_dataset(ds)
    stream = None if os.environ.get('FQL_BENCH_OWN_STREAM') else torch.cuda.current_stream().cuda_stream
    dp = DataParallelFQL(agent) if dist is not None else None
    lo, hi = shard_range(args.rows, rank, world)

    def step():
        if dp is not None:
            dp.update_from_dataset(args.rows, batch_size=B)
        else:
            agent.update_from_dataset(B, stream=stream)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the stream the graphs are launched on
    info = agent.read_info()
    st = agent.stats()

    if rank == 0:
        steps_per_s = args.steps / dt
        flop_per_step = 2.0 * st['macs_per_update']            # algorithmic: SURVEY.md 8d (12.376 GFLOP at B=256)
        step_us_dev = dev_ms * 1e3 / args.steps                # device time per update (HIP events on the launch stream)
        step_us_wall = dt * 1e6 / args.steps                   # fenced wall clock per update (>= device time: enqueue is async)
        # the events bracket the launch stream only; when the graph's side lanes outlast it the wall clock is the honest
        # duration, so the roofline is priced on the LARGER of the two
        achieved = flop_per_step / (max(step_us_dev, step_us_wall) * 1e-6) / 1e12
        out = {
            'metric': 'FQL gradient-steps/s (batch=256)', 'value': round(steps_per_s * world, 2), 'unit': 'grad-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt * 1e3 / args.steps, 5),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('visual-cube-shaped synthetic replay (uint8 64x64x9 = 3 stacked frames, act=5), impala_small encoders, '
                                    f'batch={B}/GPU, hidden=512x4, flow_steps=10, alpha=300, p_aug=0.5, {args.frames} device-resident frames '
                                    '(BASELINE.json configs[4])') if visual else
                                   (f'antmaze-large-shaped synthetic replay (obs={od}, act={ad}), batch={B}/GPU, hidden=512x4, '
                                    f'flow_steps=10, alpha={cfg["alpha"]:g}, 1M device-resident transitions (BASELINE.json configs[1])'
                                    if (od, ad, B) == (29, 8, 256) else
                                    f'synthetic replay (obs={od}, act={ad}), batch={B}/GPU, hidden=512x4, flow_steps=10, '
                                    f'alpha={cfg["alpha"]:g}, device-resident transitions (BASELINE.json configs[2] shape when 40/4/1024)'),
                       'parallelism': f'dp{world}', 'samples_per_s': round(steps_per_s * world * B, 1)},
            'roofline': {'bound': 'mfma', 'achieved': round(achieved, 3), 'peak': FP32_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved / FP32_MATRIX_PEAK_TFLOPS, 4), 'traffic': None,
                         'kernel': ('whole update graph (fql_conv3x3_kernel + fql_conv_wgrad_kernel dominate)' if visual else
                                    'whole update graph (MFMA tile kernels fql_gemm16_kernel + fql_wgrad_kernel dominate)'),
                         'flop_per_launch': flop_per_step, 'launch_us': round(step_us_dev, 3), 'wall_us': round(step_us_wall, 3),
                         'kernel_launches_per_update': st['launches_per_update']},
            'last_info': {k: round(v, 5) for k, v in info.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, od, ad, B, img=(64, 64, 9) if visual else None)
            if not visual:
                # "loss delta vs the reference" (BASELINE.json metric), against the CPU restatement: one total_loss on the
                # engine's CURRENT parameters and a fixed synthetic batch with explicit noise, fp64 oracle (checker only)
                pb = O.sample_batch(ds, np.random.default_rng(3).integers(0, args.rows, size=B))
                pn = O.make_noise(B, ad, 4)
                ref = O.OracleFQL(agent.get_params(), {k: v for k, v in dict(cfg).items() if k != 'rng'}, od, ad, np.float64)
                lg, ig = agent.total_loss(pb, None, noise=pn)
                lr, ir = ref.total_loss(pb, pn)
                out['cpu_baseline']['loss_delta'] = {'total_loss_gpu': round(float(lg), 6), 'total_loss_oracle': round(float(lr), 6),
                                                     'abs_delta': float(abs(lg - lr)),
                                                     'max_abs_delta_info': float(max(abs(ig[k] - ir[k]) for k in ir))}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
