#!/usr/bin/env python3
"""bench.py -- MCTS simulations/sec of the lockstep self-play loop on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N>1 it is
launched through torch.distributed.run, one rank per GPU.  One *step* is one lockstep
pass of the hot path: one MCTS simulation (select -> expand S1 -> tower -> reply S2 ->
tower -> backup) for EACH of the G games resident on the GPU, move boundaries (fresh
tree, root priors, host compute_policy, two pushes) included whenever a game's budget of
800 simulations completes.  ``value`` = simulations completed by all ranks / wall time of
the K timed steps (max over ranks), state resident in HBM throughout.

Workload (BASELINE.json metric "MCTS simulations/sec at 800 sims/move", config C3 --
fits one GPU): 4096 games in lockstep per GPU, 800 sims/move, 10-block/128-filter
random-init tower, fp16 MFMA trunk, all games from the standard position, Dirichlet
noise on.  Games are independent: ranks share nothing on the hot path (weak scaling).

Extra objects on the JSON line: ``roofline`` for the dominant kernel (the 3x3
residual-block convolution, MFMA-bound), ``roofline_tree`` for the hand-written HIP
search kernels (HBM-bound), ``cpu_baseline`` for the reference-shaped CPU port
(oracle/, config C1) timed on this box's host cores on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md, dense
HBM_PEAK_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1600)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--games", type=int, default=4096, help="games in lockstep per GPU")
    p.add_argument("--sims", type=int, default=800)
    p.add_argument("--blocks", type=int, default=10)
    p.add_argument("--filters", type=int, default=128)
    p.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"])
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-fused", action="store_true", help="PyTorch-ROCm trunk instead of the HIP kernel")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    return p.parse_args()


def event_time_ms(fn, reps, stream_sync=True):
    """Average duration of fn() on torch's current stream, HIP events around `reps` calls."""
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def profile_phases(run, n):
    """HIP-event time of each phase of a step, measured eagerly (no graph) on `n` real steps in
    the middle of a move, on the stream the kernels are launched on."""
    eng = run.engine
    if run._sims_in_move is not None:
        run.end_move()
    run.begin_move()
    for _ in range(max(0, run.sims // 2 - n)):
        eng.step()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(n)]
    c = eng.ctx
    for i in range(n):
        ev[i][0].record()
        c.sim_select_expand(eng.pol_s2.data_ptr(), eng.val_s2.data_ptr(), eng.planes_s1.data_ptr())
        ev[i][1].record()
        eng._eval_into(eng.planes_s1, eng.pol_s1, None)
        ev[i][2].record()
        c.sim_reply(eng.pol_s1.data_ptr(), eng.planes_s2.data_ptr())
        ev[i][3].record()
        eng._eval_into(eng.planes_s2, eng.pol_s2, eng.val_s2)
        ev[i][4].record()
    torch.cuda.synchronize()
    names = ["select_expand", "tower_s1", "reply", "tower_s2"]
    return {nm: sum(ev[i][k].elapsed_time(ev[i][k + 1]) for i in range(n)) / n
            for k, nm in enumerate(names)}


def cpu_baseline(seconds):
    """Reference-shaped CPU self-play (config C1: 1 game, 50 sims/move, tiny random-init net,
    sequential object tree, one tower call per request), bounded to ~`seconds` of CPU work."""
    from oracle import mcts_oracle, tower_oracle
    torch.set_num_threads(1)
    w = tower_oracle.init_weights(2, 32, seed=0)
    agent = mcts_oracle.OracleAgent(tower_oracle.TowerNet(w))
    from oracle.chess_oracle import OracleGame
    gam = OracleGame(player_color=True)
    sims, moves, t0 = 50, 0, time.perf_counter()
    while gam.get_result() is None and time.perf_counter() - t0 < seconds:
        bm, am = agent.best_move(gam, real_game=False, ai_move=True, max_iters=sims, noise=True)
        gam.move(bm)
        gam.move(am)
        moves += 1
    dt = time.perf_counter() - t0
    return {"value": moves * sims / dt, "unit": "simulations/s", "cores": 1, "kind": "port",
            "sample": "oracle/ (CPU restatement of selfplay.py->mctree.py + python-chess rules + fp32 "
                      "tower): 1 game, %d moves x %d sims, 2-block/32-filter net, %d tower calls, %.1f s; "
                      "reference's own figure: 0.4 s/iteration on an i5-7600K (DOCS.md:75)"
                      % (moves, sims, agent.n_evals, dt),
            "host_cores": os.cpu_count()}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # self-test hooks (1-GPU box): CRL_BENCH_DEVICE pins every rank to one device and
    # CRL_BENCH_BACKEND=gloo replaces RCCL, so the multi-rank control flow can be exercised there
    if "CRL_BENCH_DEVICE" in os.environ:
        local = int(os.environ["CRL_BENCH_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        backend = os.environ.get("CRL_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from chessrl_amd.model import ChessModel
    from chessrl_amd.selfplay import SelfPlayRunner
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[a.dtype]
    model = ChessModel(blocks=a.blocks, filters=a.filters, device="cuda:%d" % local, dtype=tdt,
                       seed=a.seed, fused=not a.no_fused)
    run = SelfPlayRunner(model, a.games, a.sims, seed=a.seed, noise=True, rank=rank, world=world,
                         device=local, use_graph=not a.no_graph, max_plies=2048)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        run.step()
    c0 = run.engine.ctx.counters()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    c1 = run.engine.ctx.counters()
    dt = t1 - t0
    sims = c1["sims"] - c0["sims"]          # simulations completed (backed up) in the timed region
    tot = torch.tensor([float(sims), dt], dtype=torch.float64, device=dev)
    if world > 1:
        s = tot.clone()
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        m = tot.clone()
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        total_sims, max_dt = s[0].item(), m[1].item()
    else:
        total_sims, max_dt = float(sims), dt

    out = None
    if rank == 0:
        G, F, B = a.games, a.filters, a.blocks
        eng = run.engine
        d = {k: c1[k] - c0[k] for k in c1}
        nodes = max(1, d["nodes"])
        depth = d["depth_sum"] / max(1, d["sims"])
        branch = d["branch_sum"] / nodes
        # ---- dominant kernel (MFMA-bound) ----------------------------------------------------
        tower_ms = event_time_ms(lambda: model.forward_into(eng.planes_s2, eng.pol_s2, eng.val_s2), 20)
        tower_flops = 2.0 * model.macs_per_eval() * G
        peak = MFMA_PEAK_TFLOPS[a.dtype]
        if model.fused:
            # crl_tower::k_trunk128 -- the hand-written fused trunk: ONE launch per tower forward.
            # Algorithmic FLOPs per launch = 2 x (stem 73152 F + blocks 1152 F^2 B + head convs
            # 192 F) MACs per board x G boards.
            k_ms = event_time_ms(lambda: model._run_fused(eng.planes_s2), 50)
            k_flops = 2.0 * (73152 * F + 1152 * F * F * B + 192 * F) * G
            k_name = "crl_tower::%s (fused stem + %d residual blocks + head convs, %d boards)" % (
                # the dispatch rule of crl_trunk_forward (csrc/api.hip)
                ("k_trunk_gen<%d, %d, %d>" % (F, (1 if F == 256 else 2), eng.bitplanes)
                 if G <= 128 * (2 if F == 256 else 4)
                 else "k_trunk128_pipe<0, %d>" % eng.bitplanes if F == 128
                 else "k_trunk_gen<%d, %d, %d>" % (F, 2 if F == 256 else 4, eng.bitplanes)),
                B, G)
        else:
            # PyTorch-ROCm tower: the FxF 3x3 residual-block convolution.  bias=None: exactly ONE
            # kernel per call (MIOpen igemm_fwd_gtcx35_nhwc_*), comparable with rocprofv3 --stats
            conv = model.net.conv1[0]
            x = torch.randn((G, F, 8, 8), device=dev, dtype=tdt).contiguous(memory_format=torch.channels_last)
            k_ms = event_time_ms(lambda: torch.nn.functional.conv2d(x, conv.weight, None, padding=1), 50)
            k_flops = 2.0 * 9 * F * F * 64 * G
            k_name = "3x3 conv %d->%d, batch %d x 8x8 (PyTorch-ROCm / MIOpen igemm)" % (F, F, G)
        # traffic: rocprofv3 PMC passes of this kernel at this exact shape (profiles/r01/
        # pmc_trunk_kernel.md): FETCH_SIZE 129 858 KB x 2 (gfx950 correction) + WRITE_SIZE 3 072 KB
        traffic = 2 * 129858e3 + 3072e3 if (model.fused and (G, F, B) == (4096, 128, 10)) else None
        roof = {"bound": "mfma", "kernel": k_name,
                "achieved": k_flops / k_ms / 1e9, "peak": peak, "unit": "TFLOP/s",
                "frac": k_flops / k_ms / 1e9 / peak, "traffic": traffic,
                "algorithmic_bytes": (G * 64 * 128 * 2 + 9 * 128 * F * 2 + 2 * B * 9 * F * F * 2 + G * 192 * 4)
                if model.fused else None,
                "launch_ms": k_ms, "flops_per_launch": k_flops,
                "tower_forward_ms": tower_ms, "tower_tflops": tower_flops / tower_ms / 1e9,
                "tower_frac": tower_flops / tower_ms / 1e9 / peak}
        # ---- hand-written HIP search kernels (HBM-bound): select+expand and reply -----------
        # algorithmic bytes per simulation, SURVEY.md section 8d with the measured d and b;
        # with the fused trunk the encoders hand over 1-KiB plane bitboards (expanded on chip); any
        # other evaluator gets 16-KiB fp16 planes.  Policy vectors are materialised (+2*b*4 B gathered)
        plane_bytes = 1024.0 if eng.bitplanes else 16384.0      # per evaluated position (S1 and S2)
        tree_bytes = 2400.0 + 20.0 * depth * branch + 2 * plane_bytes
        ph = profile_phases(run, 16)
        tree_ms = ph["select_expand"] + ph["reply"]
        tree = {"bound": "hbm", "kernel": "k_select_expand + k_reply (one simulation x %d games)" % G,
                "achieved": tree_bytes * G / tree_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": tree_bytes * G / tree_ms / 1e6 / HBM_PEAK_GBS,
                # PMC passes of both kernels at 4096 games (profiles/r01/pmc_tree_kernels.md): bytes/step
                "traffic": ((2 * (16054e3 + 6415e3) + 42729e3 + 7731e3) if eng.bitplanes else
                            (2 * (27370e3 + 6435e3) + 102267e3 + 69172e3)) if G == 4096 else None,
                "launch_ms": tree_ms, "phase_ms": ph, "bytes_per_sim": tree_bytes,
                "mean_depth": depth, "mean_branch": branch}
        out = {
            "metric": "MCTS simulations/sec at 800 sims/move", "value": total_sims / max_dt,
            "unit": "simulations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": max_dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "%s: %d self-play games in lockstep per GPU, %d sims/move, "
                                   "%d-block/%d-filter random-init tower, standard start position, "
                                   "Dirichlet noise on" % (
                                       {(512, 100, 6, 64): "C2", (4096, 800, 10, 128): "C3 (= C4 per-GPU shard)",
                                        (4096, 800, 20, 256): "C5 per-GPU shard"}.get(
                                           (G, a.sims, B, F), "custom"), G, a.sims, B, F),
                       "games_per_gpu": G, "sims_per_move": a.sims, "tower": "%dx%d" % (B, F),
                       "hipgraph": not a.no_graph, "fused_trunk_kernel": bool(model.fused), "parallelism": "games sharded, no collective on the hot path"},
            "moves_per_sec": total_sims / max_dt / a.sims,
            # games/hour: a random-init 10x128 net at 800 sims/move plays 165.5 moves (331 plies) per
            # game on average (512 complete games, profiles/r01/game_length_c3net_800sims.json;
            # 190 moves at 50 sims/move); steady state with refill = moves/s / moves per game
            "self_play_games_per_hour_est": total_sims / max_dt / a.sims / 165.5 * 3600.0,
            "tower_evals_per_sim": d["evals"] / max(1, d["sims"]),
            "gflop_per_sim": 2 * 2 * model.macs_per_eval() / 1e9,
            "roofline": roof, "roofline_tree": tree,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()            # rank 0 is still profiling its phases: nobody tears NCCL down early
    run.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
