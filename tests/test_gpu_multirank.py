"""Two ranks through the HIP path (run with ``-m gpu`` on the one-GPU box).

SURVEY 8(e): the batch shards by contiguous lane ranges, global game ids feed the Philox counters, and the
only collective is a SUM all-reduce of the counters.  Here two fresh child processes (``torch.distributed.run``,
backend gloo, both on ``cuda:0``, launched before anything in them touches the GPU) each step their shard through
``pikazoo_amd``; the concatenated shards must equal the single-process batch bit for bit, and the all-reduced
counters must equal its totals.  The same launcher then runs ``bench.py --gpus 2`` the way the driver does (with
gloo standing in for RCCL on a one-GPU box) and checks the line it prints -- and the one-rank line of the driver's own
command: flat, below 8 KB, one scalar per BASELINE-config figure.
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parent.parent

_WORKER = r'''
import os, sys
repo, out_dir, n_global, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
sys.path.insert(0, os.path.join(repo, "pika-zoo_amd")); sys.path.insert(0, repo)
import torch
from pikazoo_amd import dist, pikazoo_v0

rank, world, _ = dist.init_from_env("gloo")          # before any GPU call in this process
assert world == 2 and dist.world_size() == 2 and dist.backend_name() == "gloo"
shard = dist.shard_for_rank(n_global, rank, world)
results = {}
for name, kw in (("human", {}), ("cfg3", dict(is_player2_computer=True)), ("both", dict(is_player1_computer=True, is_player2_computer=True))):
    env = pikazoo_v0.env(num_envs=shard.n_local, device="cuda:0", seed=21, env_id_base=shard.env_id_base,
                         winning_score=1, validate_actions=False, **kw)
    env.reset()
    terms = 0
    for t in range(steps):                            # one pz_step launch per frame (pair kernel)
        obs, rew, term, _, _ = env.step(env.random_actions(8, t))
        terms += int(term["player_1"].sum().item())
    env.step_random(8, t0=steps, k=64)                # and one k-frame launch
    torch.cuda.synchronize()
    total_terms, total_envs, total_eps = dist.all_reduce_sum([terms, shard.n_local, env.episodes_done])
    results[name] = {"state": env.state.cpu(), "obs1": obs["player_1"].cpu(), "terms": terms,
                     "total_terms": total_terms, "total_envs": total_envs, "total_eps": total_eps,
                     "base": shard.env_id_base}
torch.save(results, os.path.join(out_dir, f"rank{rank}.pt"))
dist.barrier()
torch.distributed.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(args, timeout=420):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=str(REPO))


def test_two_ranks_through_the_hip_path_equal_the_single_batch(tmp_path):
    from pikazoo_amd import pikazoo_v0

    n_global, steps = 8192 + 200, 150  # ragged: the second shard starts inside a 64-game workgroup span
    worker = tmp_path / "worker.py"
    worker.write_text(_WORKER)
    r = _launch([str(worker), str(REPO), str(tmp_path), str(n_global), str(steps)])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    shards = [torch.load(tmp_path / f"rank{k}.pt") for k in range(2)]
    for name, kw in (("human", {}), ("cfg3", dict(is_player2_computer=True)),
                     ("both", dict(is_player1_computer=True, is_player2_computer=True))):
        env = pikazoo_v0.env(num_envs=n_global, device="cuda:0", seed=21, env_id_base=0, winning_score=1,
                             validate_actions=False, **kw)
        env.reset()
        terms = 0
        for t in range(steps):
            obs, rew, term, _, _ = env.step(env.random_actions(8, t))
            terms += int(term["player_1"].sum().item())
        env.step_random(8, t0=steps, k=64)
        a, b = shards[0][name], shards[1][name]
        assert a["base"] == 0 and b["base"] == a["state"].shape[1]
        assert torch.equal(torch.cat([a["state"], b["state"]], dim=1), env.state.cpu()), name
        assert torch.equal(torch.cat([a["obs1"], b["obs1"]], dim=0), obs["player_1"].cpu()), name
        # the one collective: counters summed over ranks == the single batch's totals, on both ranks
        for sh in (a, b):
            assert sh["total_envs"] == n_global and sh["total_terms"] == terms == a["terms"] + b["terms"], name
            assert sh["total_eps"] == env.episodes_done, name
        assert terms > 0


def test_bench_with_two_ranks_prints_the_aggregate_line():
    """Plain `python bench.py --gpus 2`: bench.py starts its two ranks itself (a child torch.distributed.run, what the
    driver would have started); both share cuda:0 (PZ_BENCH_ONE_DEVICE) and gloo stands in for RCCL.  value must be
    the whole-job aggregate, every rank reports its own figures, and the first and the last rank replay their games
    on the oracle."""
    env = dict(os.environ, PZ_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "64", "--warmup", "8", "--burn-in", "128",
           "--min-time", "0.05", "--dist-backend", "gloo", "--cpu-threads", "4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(REPO))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    # gloo carried the counters: that is said, and not counted as RCCL ranks
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["rccl_ranks"] == 0 and out["dist_backend"] == "gloo"
    assert out["config"]["num_envs_total"] == 2 * 65536 and out["scaling"] == "weak"
    assert out["timed_steps"] >= 64 and out["timed_seconds"] >= 0.05
    per_step_s = out["ms_per_step"] * 1e-3
    assert np.isclose(out["value"], 2 * 65536 / per_step_s, rtol=1e-6)
    assert "configs" not in out and "cpu_baseline" not in out  # single-GPU extras stay off for N > 1
    assert [p["rank"] for p in out["per_rank"]] == [0, 1]
    for p in out["per_rank"]:
        assert p["launch_us"] > 0 and p["value"] > 0 and p["parity_bit_exact"] is True  # both ends of the job checked


def test_bench_with_four_ranks_on_one_device_shards_like_config_4():
    """`python bench.py --gpus 4 --num-envs 65536` -- BASELINE config 4's per-rank shard at the largest rank count the
    one-GPU box allows (six processes may have its card open: four ranks, their launcher and this one -- five ranks were
    killed by the box's process guard; the eight-rank integers are covered over gloo on the CPU,
    tests/test_cabi_and_host.py): `per_rank` has a row per rank, rank r's games start at r * 65 536 (the first and the
    last rank replay theirs on the oracle under those ids), the job counts 4 x 65 536 games."""
    ranks = 4
    env = dict(os.environ, PZ_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(REPO / "bench.py"), "--gpus", str(ranks), "--num-envs", "65536", "--steps", "16", "--warmup", "4",
           "--burn-in", "64", "--min-time", "0.02", "--dist-backend", "gloo", "--cpu-threads", "2", "--no-configs",
           "--check-lanes", "256"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(REPO))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == ranks and out["ranks"] == ranks and out["rccl_ranks"] == 0 and out["scaling"] == "weak"
    assert out["config"]["num_envs_total"] == ranks * 65536 and out["config"]["num_envs_per_gpu"] == 65536
    assert [p["rank"] for p in out["per_rank"]] == list(range(ranks))
    assert [p["env_id_base"] for p in out["per_rank"]] == [k * 65536 for k in range(ranks)]
    assert out["per_rank"][0]["parity_bit_exact"] is True and out["per_rank"][-1]["parity_bit_exact"] is True
    assert all(p["parity_bit_exact"] is None for p in out["per_rank"][1:-1])
    assert np.isclose(out["value"], ranks * 65536 / (out["ms_per_step"] * 1e-3), rtol=1e-6)
    # the N-rank story once more as FLAT scalars of `config` / `roofline`: a record that keeps scalar members only (the
    # driver's) still says who ran, over what, how the slowest rank did and that both ends of the job were oracle-checked
    cfg, roof = out["config"], out["roofline"]
    assert cfg["ranks"] == ranks and cfg["rccl_ranks"] == 0 and cfg["dist_backend"] == "gloo" and isinstance(cfg["dist_note"], str)
    assert cfg["env_id_base_last_rank"] == (ranks - 1) * 65536 and len(cfg["build_id"]) == 16
    assert roof["per_rank_min_value"] == min(p["value"] for p in out["per_rank"]) > 0
    assert roof["per_rank_max_launch_us"] == max(p["launch_us"] for p in out["per_rank"]) > 0
    assert roof["parity_first_last_rank_bit_exact"] is True and roof["parity_ranks_checked"] == 2
    assert all(not isinstance(v, (dict, list)) for v in list(cfg.values()) + list(roof.values()))
    assert len(json.dumps(out)) < 8000


def test_bench_under_the_drivers_launcher_and_rccl_request_on_one_gpu():
    """The driver's own launch line (torch.distributed.run around bench.py), asking for RCCL: two ranks on one GPU is
    what RCCL refuses, so all ranks agree -- before any of them touches RCCL -- to leave the counters on gloo, and the
    line says so instead of claiming two RCCL ranks."""
    env = dict(os.environ, PZ_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(REPO / "bench.py"), "--gpus", "2", "--steps", "32",
           "--warmup", "4", "--burn-in", "64", "--min-time", "0.05", "--no-cpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(REPO))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["ranks"] == 2 and out["rccl_ranks"] == 0 and out["dist_backend"] == "gloo"
    assert "share one GPU" in out["dist_note"]
    assert all(p["parity_bit_exact"] is None for p in out["per_rank"])  # --no-cpu: nothing was replayed
    assert out["roofline"]["parity_first_last_rank_bit_exact"] is None and out["roofline"]["parity_ranks_checked"] == 0
    assert out["config"]["rccl_ranks"] == 0 and "share one GPU" in out["config"]["dist_note"]


def test_the_default_bench_line_is_flat_below_8_kb_and_carries_every_baseline_config(tmp_path):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -- the driver's own command.  Its record keeps the scalar members
    of `config` / `roofline` / `cpu_baseline` and drops nested objects, and its stdout tail is 8 KB: so the line must be
    flat, short, and carry one scalar per figure the documents quote for a BASELINE config -- measured on the
    cold action tape whatever --steps is, with the launch-floor twin of the diagnostics library and the two-chain
    measurement beside the headline, every oracle replay of the run bit-exact, and the verbose blocks in the side file."""
    side = tmp_path / "configs.json"
    cmd = [sys.executable, str(REPO / "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2",
           "--configs-out", str(side)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(REPO))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8000
    out = json.loads(lines[0])
    for block in ("config", "roofline", "cpu_baseline"):
        nested = {k: v for k, v in out[block].items() if isinstance(v, (dict, list))}
        assert not nested, (block, list(nested))
        assert all(not isinstance(v, str) or len(v) <= 128 for v in out[block].values()), block  # (the record cuts strings at 128)
    cfg, roof, cpu = out["config"], out["roofline"], out["cpu_baseline"]
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["warmup"] == 5 and out["unit"] == "env-steps/s"
    assert out["dtype"] == "int32" and out["scaling"] == "weak" and out["vs_baseline"] is None and out["higher_is_better"] is True
    assert np.isclose(out["value"], 65536 / (out["ms_per_step"] * 1e-3), rtol=1e-6)
    assert cfg["action_tape"] == "cold" and cfg["action_tape_bytes"] > 1 << 30 and cfg["num_envs_per_gpu"] == 65536
    assert len(cfg["build_id"]) == 16 and cfg["ranks"] == 1 and "workload" in cfg
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s"
    assert np.isclose(roof["frac"], roof["achieved"] / roof["peak"]) and 0.5 < roof["frac"] < 1.0
    assert np.isclose(roof["achieved"], 649 * 65536 / (roof["launch_us"] * 1e-6) / 1e9, rtol=1e-6)
    # the counted bytes: either of this build (or of a build with the same kernel), or declared stale and absent
    assert (roof["traffic_stale"] is False and roof["traffic"] > 0 and len(roof["traffic_build_id"]) == 16) or \
           (roof["traffic_stale"] is True and roof["traffic"] is None and roof["frac_traffic"] is None)
    want = ["cold_tape_launch_us", "cold_tape_frac", "hot_tape_launch_us", "hot_tape_frac", "floor_empty_us", "floor_loads_us",
            "floor_loads_stores_us", "floor_stand_in_us", "cfg2_launch_us", "cfg2_frac", "cfg3_launch_us", "cfg3_frac",
            "cfg3_compute_launch_us", "cfg5_launch_us", "cfg5_frac", "n524288_launch_us", "n524288_frac", "packed_launch_us",
            "packed_524288_launch_us", "rollout_k32_us_per_frame", "rollout_k32_frac", "step_many_k32_us_per_frame",
            "rollout_k32_p2_computer_us_per_frame", "step_many_k32_p2_computer_us_per_frame", "rollout_k128_us_per_frame",
            "rollout_k32_4096_us_per_frame", "policy_fused_launch_us", "two_chains_one_graph_us", "two_graphs_two_streams_us",
            "per_rank_min_value", "per_rank_max_launch_us"]
    missing = [k for k in want if not isinstance(roof.get(k), float)]
    assert not missing, missing
    assert roof["cold_tape_launch_us"] == round(roof["launch_us"], 4) and roof["hot_tape_launch_us"] < roof["cold_tape_launch_us"] * 1.02
    assert roof["floor_empty_us"] < roof["floor_loads_us"] < roof["floor_loads_stores_us"] < roof["floor_stand_in_us"] * 1.05
    assert roof["configs_parity_all_bit_exact"] is True and roof["configs_parity_checked"] >= 18
    assert roof["two_chains_parity_bit_exact"] is True and roof["parity_first_last_rank_bit_exact"] is True
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["parity_bit_exact"] is True
    assert cpu["reference_python_steps_per_s_per_core"] == 55100 and "sample" in cpu
    verbose = json.loads(side.read_text())
    assert {"configs", "per_rank", "regimes", "launch_floor", "other_action_tape", "traffic_status"} <= set(verbose)
    assert verbose["configs"]["cfg3"]["parity_bit_exact"] is True and out["configs_file"] == str(side)
