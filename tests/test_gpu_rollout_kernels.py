"""GPU tests of the rollout bookkeeping kernels (csrc/rollout_kernels.hip) against the composed torch ops that restate
A2CBase.play_steps / discount_values (lib/agent/a2c_base.py:463-478, 651-695) and against the Philox oracle."""
import ctypes
import math

import numpy as np
import pytest
import torch

from oracle import philox

pytestmark = pytest.mark.gpu


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture(scope="module")
def lib():
    from airgym_amd import _native as N
    assert torch.cuda.is_available()
    return N.load()


@pytest.mark.parametrize("A,normalize_value", [(4, True), (5, False), (4, False)])
def test_policy_sample(lib, A, normalize_value):
    from airgym_amd import _native as N
    from airgym_amd.lib.model.a2c_continuous_logstd_model import ModelA2CContinuousLogStd
    n, H, slot, seed, offset = 3001, 24, 7, 0x1234567887654321, 1000
    g = torch.Generator(device="cuda").manual_seed(0)
    heads = torch.randn(n, A + 1, device="cuda", generator=g)
    heads[:, A] *= 4                                   # some |v| > 5 so the de-normalisation clamp is exercised
    logstd = 0.3 * torch.randn(A, device="cuda", generator=g)
    vmean = torch.tensor([0.7], dtype=torch.float64, device="cuda")
    vvar = torch.tensor([2.5], dtype=torch.float64, device="cuda")
    counter = torch.tensor([5], dtype=torch.int64, device="cuda")
    f = dict(device="cuda", dtype=torch.float32)
    actions, mus, sigmas, env_a = (torch.empty(n, A, **f) for _ in range(4))
    nlp, values = torch.empty(n, **f), torch.empty(n, **f)
    N.check(lib.ag_policy_sample(heads.data_ptr(), logstd.data_ptr(), vmean.data_ptr() if normalize_value else None,
                                 vvar.data_ptr() if normalize_value else None, 1e-5, seed, counter.data_ptr(), H, slot, offset,
                                 actions.data_ptr(), nlp.data_ptr(), values.data_ptr(), mus.data_ptr(), sigmas.data_ptr(),
                                 env_a.data_ptr(), n, A, _stream()), "ag_policy_sample")
    mu, sigma = heads[:, :A], torch.exp(logstd).expand(n, A)
    assert torch.equal(mus, mu) and torch.allclose(sigmas, sigma, rtol=1e-6)
    # the noise is the oracle's Philox / Box-Muller stream 16 at tick = counter * H + slot
    z_ref = philox.normals(seed, np.arange(offset, offset + n), 5 * H + slot, 16, A)[:, :A]
    z = ((actions - mu) / sigma).cpu().numpy()
    np.testing.assert_allclose(z, z_ref, rtol=0, atol=2e-4)
    ref_nlp = ModelA2CContinuousLogStd.neglogp(actions, mu, sigma, logstd.expand(n, A))
    assert torch.allclose(nlp, ref_nlp, rtol=1e-5, atol=1e-5)
    v = heads[:, A]
    ref_v = torch.sqrt(vvar.float() + 1e-5) * torch.clamp(v, -5, 5) + vmean.float() if normalize_value else v
    assert torch.allclose(values, ref_v, rtol=1e-6, atol=1e-6)
    assert torch.equal(env_a, actions.clamp(-1, 1)) and (actions.abs() > 1).any()
    # a different counter value -> different noise; the same -> identical (pure function of the counters)
    a2 = torch.empty_like(actions)
    counter.add_(1)
    N.check(lib.ag_policy_sample(heads.data_ptr(), logstd.data_ptr(), None, None, 0.0, seed, counter.data_ptr(), H, slot, offset,
                                 a2.data_ptr(), nlp.data_ptr(), values.data_ptr(), mus.data_ptr(), sigmas.data_ptr(), None,
                                 n, A, _stream()), "ag_policy_sample")
    assert not torch.equal(a2, actions)
    zz = ((a2 - mu) / sigma)
    assert abs(zz.mean().item()) < 0.05 and abs(zz.std().item() - 1.0) < 0.05


@pytest.mark.parametrize("bootstrap", [False, True])
def test_rollout_account(lib, bootstrap):
    from airgym_amd import _native as N
    n = 70001
    g = torch.Generator(device="cuda").manual_seed(1)
    raw = torch.randn(n, device="cuda", generator=g)
    dones = (torch.rand(n, device="cuda", generator=g) < 0.1).to(torch.uint8)
    tmo = (torch.rand(n, device="cuda", generator=g) < 0.2).to(torch.uint8)
    values = torch.randn(n, device="cuda", generator=g)
    cr, cs, cl = (torch.rand(n, device="cuda", generator=g) * 10 for _ in range(3))
    cr0, cs0, cl0 = cr.clone(), cs.clone(), cl.clone()
    shaped = torch.empty(n, device="cuda")
    parts = torch.zeros(lib.ag_rollout_account_blocks(n), 4, dtype=torch.float64, device="cuda")
    scale, shift, lo, hi, gamma = 0.5, 0.1, -0.9, 0.8, 0.99
    N.check(lib.ag_rollout_account(raw.data_ptr(), dones.data_ptr(), tmo.data_ptr() if bootstrap else None,
                                   values.data_ptr() if bootstrap else None, scale, shift, lo, hi, 0, gamma, shaped.data_ptr(),
                                   cr.data_ptr(), cs.data_ptr(), cl.data_ptr(), parts.data_ptr(), n, _stream()),
            "ag_rollout_account")
    ref_sh = torch.clamp((raw + shift) * scale, lo, hi)
    if bootstrap:
        ref_sh = ref_sh + gamma * values * tmo.float()
    assert torch.allclose(shaped, ref_sh, rtol=1e-6, atol=1e-7)
    r_cr, r_cs, r_cl = cr0 + raw, cs0 + ref_sh, cl0 + 1
    d = dones.bool()
    ref = torch.stack((d.double().sum(), r_cr[d].double().sum(), r_cs[d].double().sum(), r_cl[d].double().sum()))
    assert torch.allclose(parts.sum(0), ref, rtol=1e-9)
    assert torch.allclose(cr, r_cr * (~d), atol=1e-6) and torch.allclose(cs, r_cs * (~d), atol=1e-6) and torch.allclose(cl, r_cl * (~d))
    # unbounded shaper == identity clamp
    N.check(lib.ag_rollout_account(raw.data_ptr(), dones.data_ptr(), None, None, 1.0, 0.0, -math.inf, math.inf, 0, gamma,
                                   shaped.data_ptr(), cr.data_ptr(), cs.data_ptr(), cl.data_ptr(), parts.data_ptr(), n,
                                   _stream()), "ag_rollout_account")
    assert torch.equal(shaped, raw)


def test_gae_kernel(lib):
    from airgym_amd import _native as N
    from airgym_amd.lib.agent.a2c_continuous import discount_values
    H, n, gamma, tau = 24, 5003, 0.99, 0.95
    g = torch.Generator(device="cuda").manual_seed(2)
    rewards = torch.randn(H, n, 1, device="cuda", generator=g)
    values = torch.randn(H, n, 1, device="cuda", generator=g)
    dones = (torch.rand(H + 1, n, device="cuda", generator=g) < 0.1).to(torch.uint8)
    last_values = torch.randn(n, 1, device="cuda", generator=g)
    advs, rets = torch.empty_like(values), torch.empty_like(values)
    N.check(lib.ag_gae(rewards.data_ptr(), values.data_ptr(), dones.data_ptr(), last_values.data_ptr(), gamma, tau,
                       advs.data_ptr(), rets.data_ptr(), H, n, _stream()), "ag_gae")
    ref = discount_values(dones[H].float(), last_values, dones[:H].float(), values, rewards, gamma, tau)
    assert torch.allclose(advs, ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(rets, ref + values, rtol=1e-5, atol=1e-5)


def test_fused_rollout_trains(lib):
    """The six-launch rollout step feeds the same buffers as the eager path: finite training, per-env episode accounting
    consistent with the env's own progress counters, fresh noise on every hipGraph replay."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent

    class Args:
        envs = 4096; minibatches = 4; graph = 1
    agent = A2CAgent("t", bench.build_params(Args, 1))
    agent.init_tensors()
    assert agent._fused_rollout is not None and agent._fused_step is not None
    agent.obs = agent.env_reset()
    H = agent.horizon_length
    acts = []
    for ep in range(3):                      # epoch 0 eager, epoch 1 captures, epoch 2 replays the graph
        agent.epoch_num += 1
        st = agent.train_epoch()
        acts.append(agent.actions_buf.clone())
        assert all(math.isfinite(st[k]) for k in ("a_loss", "c_loss", "kl", "entropy"))
        z = (agent.actions_buf - agent.mus_buf) / agent.sigmas_buf
        assert abs(z.mean().item()) < 0.02 and abs(z.std().item() - 1) < 0.02
        # stored neglogp is what the model recomputes from the stored (action, mu, sigma)
        nlp = 0.5 * (z ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * 4 + torch.log(agent.sigmas_buf).sum(-1)
        assert torch.allclose(nlp, agent.neglogpacs_buf, rtol=1e-4, atol=1e-4)
        # running episode length == the env's progress counter for envs that did not just reset
        prog = agent._hip_env.get_state()["progress"].float()
        alive = agent.dones_buf[H] == 0
        diff = prog[alive] - agent.current_lengths[alive]      # the env counts the step that follows its reset as well
        assert ((diff == 0) | (diff == 1)).all()
    assert not torch.equal(acts[1], acts[2])
    noise1 = (acts[1] - agent.mus_buf).abs().sum()
    assert noise1 > 0
    assert agent.game_lengths.current_size > 0 or agent.ep_stats[:, 0].sum() >= 0


@pytest.mark.parametrize("rows,D", [(196608, 18), (4097, 1), (70001, 48), (2, 18)])
def test_rms_update_kernel(lib, rows, D):
    """ag_rms_update == RunningMeanStd.update's torch formulas (running_mean_std.py:31-62), twice in a row."""
    from airgym_amd.lib.core.running_mean_std import RunningMeanStd
    g = torch.Generator(device="cuda").manual_seed(4)
    hip, ref = RunningMeanStd((D,)).cuda(), RunningMeanStd((D,)).cuda()
    for it in range(2):
        x = (3 * torch.randn(rows, D, device="cuda", generator=g) + 7 * it).contiguous()
        hip.update(x)                                      # HIP path (CUDA, contiguous f32, no group)
        # reference formulas
        mean, var, n = x.mean(0).double(), x.var(0).double(), float(rows)
        delta = mean - ref.running_mean
        tot = ref.count + n
        m2 = ref.running_var * ref.count + var * n + delta ** 2 * ref.count * n / tot
        ref.running_mean.copy_(ref.running_mean + delta * n / tot)
        ref.running_var.copy_(m2 / tot)
        ref.count.copy_(tot)
        assert torch.allclose(hip.running_mean, ref.running_mean, rtol=1e-6, atol=1e-6)
        assert torch.allclose(hip.running_var, ref.running_var, rtol=2e-5, atol=1e-6)
        assert hip.count.item() == ref.count.item()


@pytest.mark.parametrize("C", [64, 128, 256])
@pytest.mark.parametrize("A1,write_back", [(5, 1), (6, 0)])
def test_elu_heads_kernel(lib, C, A1, write_back):
    """ag_elu_heads == ELU then the [M,C]x[C,A1] head product (DPP row reduction over 16 / 32 / 64 lanes)."""
    import torch.nn.functional as F
    from airgym_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(5)
    for M in (1, 63, 5001):
        z = torch.randn(M, C, device="cuda", generator=g) * 2
        Wh = torch.randn(A1, C, device="cuda", generator=g) * 0.1
        bh = torch.randn(A1, device="cuda", generator=g)
        heads = torch.empty(M, A1, device="cuda")
        buf = z.clone()
        zb = torch.randn(C, device="cuda", generator=g) if write_back == 0 else None
        N.check(lib.ag_elu_heads(buf.data_ptr(), Wh.data_ptr(), bh.data_ptr(), heads.data_ptr(), M, C, A1, write_back,
                                 zb.data_ptr() if zb is not None else None, _stream()),
                "ag_elu_heads")
        h = F.elu(z + zb) if zb is not None else F.elu(z)
        assert torch.allclose(heads, h @ Wh.t() + bh, rtol=1e-5, atol=1e-5)
        if write_back:
            assert torch.allclose(buf, h, rtol=1e-6, atol=1e-6)
        else:
            assert torch.equal(buf, z)


def test_sum_rows_multi(lib):
    """ag_sum_rows_multi: several [rows, n] partial tables reduced over rows in two launches, into unaligned outputs."""
    from airgym_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(6)
    shapes = [(1536, 1280), (1536, 256), (64, 65536), (768, 256), (768, 4608), (1, 8), (7, 12), (3000, 4)]
    flat = torch.zeros(sum(n for _, n in shapes) + 1, device="cuda")
    parts, outs, off = [], [], 1                     # offset 1: destinations only 4-byte aligned
    for rows, n in shapes:
        parts.append(torch.randn(rows, n, device="cuda", generator=g))
        outs.append(flat[off:off + n]); off += n
    jobs = (N.AgSumJob * len(shapes))()
    for j, (p, o) in enumerate(zip(parts, outs)):
        jobs[j] = N.AgSumJob(p.data_ptr(), o.data_ptr(), p.shape[0], p.shape[1])
    scratch = torch.empty(lib.ag_sum_rows_groups() * sum(n for _, n in shapes), device="cuda")
    for _ in range(2):
        N.check(lib.ag_sum_rows_multi(jobs, len(shapes), scratch.data_ptr(), scratch.numel(), _stream()), "ag_sum_rows_multi")
    for p, o in zip(parts, outs):
        ref = p.double().sum(0)
        assert torch.allclose(o.double(), ref, rtol=1e-5, atol=1e-4 * max(1.0, p.shape[0] ** 0.5))
    assert flat[0] == 0
    first = flat.clone()
    N.check(lib.ag_sum_rows_multi(jobs, len(shapes), scratch.data_ptr(), scratch.numel(), _stream()), "ag_sum_rows_multi")
    assert torch.equal(first, flat), "fixed summation order: bit-identical on repeat"


def test_sum_rows_multi_with_the_loss_finalize_riding_along(lib):
    """ag_sum_rows_multi_finalize == ag_ppo_loss_finalize + ag_sum_rows_multi, bit for bit (one launch less per optimizer step)."""
    from airgym_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(16)
    A, M, nb = 4, 196608, 768
    ns = lib.ag_ppo_loss_num_sums()
    lp = torch.randn(nb, ns, device="cuda", generator=g)
    logstd = 0.1 * torch.randn(A, device="cuda", generator=g)
    shapes = [(768, 1280), (768, 256), (256, 65536), (768, 256), (768, 4608)]
    parts = [torch.randn(r, n, device="cuda", generator=g) for r, n in shapes]
    scratch = torch.empty(lib.ag_sum_rows_groups() * sum(n for _, n in shapes), device="cuda")
    res = []
    for fused in (False, True):
        outs = [torch.zeros(n, device="cuda") for _, n in shapes]
        jobs = (N.AgSumJob * len(shapes))(*[N.AgSumJob(p.data_ptr(), o.data_ptr(), p.shape[0], p.shape[1]) for p, o in zip(parts, outs)])
        gl, gb, kl, st = (torch.full((k,), 7.0, device="cuda") for k in (A, A + 1, 1, 8))
        fin = (lp.data_ptr(), nb, M, A, logstd.data_ptr(), 0.01, 2.0, 1e-4, gl.data_ptr(), gb.data_ptr(), kl.data_ptr(), st.data_ptr())
        if fused:
            N.check(lib.ag_sum_rows_multi_finalize(jobs, len(shapes), scratch.data_ptr(), scratch.numel(), *fin, _stream()), "fused")
        else:
            N.check(lib.ag_ppo_loss_finalize(*fin, _stream()), "finalize")
            N.check(lib.ag_sum_rows_multi(jobs, len(shapes), scratch.data_ptr(), scratch.numel(), _stream()), "sum")
        torch.cuda.synchronize()
        res.append(outs + [gl, gb, kl, st])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert (res[1][-1][:7] != 7.0).all() and res[1][-1][7] == 0.0            # the stats row was written
    jobs = (N.AgSumJob * 1)(N.AgSumJob(parts[0].data_ptr(), res[0][0].data_ptr(), 768, 1280))
    assert lib.ag_sum_rows_multi_finalize(jobs, 1, scratch.data_ptr(), scratch.numel(), None, nb, M, A, logstd.data_ptr(), 0.0, 1.0, 0.0,
                                          res[0][5].data_ptr(), res[0][6].data_ptr(), res[0][7].data_ptr(), res[0][8].data_ptr(),
                                          _stream()) == -1


def test_minibatch_graphs_are_bit_identical_to_eager(lib):
    """use_hip_graph: rollout graph + one captured graph per minibatch step == the eager launch sequence, bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    results = []
    # (graph, minibatch graphs): everything captured | rollout graph with an eager update (the headline's regime: 196 608-row
    # minibatches are not launch-bound) | everything eager
    for graph, upd in ((1, True), (1, False), (0, False)):
        class Args:
            envs = 2048; minibatches = 8; task = "hovering"; ctl = "rate"; tuned_gemms = 1
        Args.graph = graph
        torch.manual_seed(0)
        params = bench.build_params(Args, 1)
        if graph and not upd:
            params["config"]["use_hip_graph_update"] = False
        agent = A2CAgent("g", params)
        assert agent._graph_update == upd
        agent.init_tensors()
        agent.obs = agent.env_reset()
        for ep in range(1, 5):
            agent.epoch_num = ep
            st = agent.train_epoch()
        if graph:
            assert len(agent._upd_graphs) == (16 if upd else 0)      # 8 minibatches x {stats on, off}
            assert "rollout" in agent._graphs
        else:
            assert not agent._graphs
        results.append((agent.flat_param.clone(), agent.optimizer.lr.item(), st["kl"],
                        agent.model.running_mean_std.running_mean.clone(), agent.value_mean_std.running_var.clone(),
                        agent.game_rewards.get_mean().copy(), st["a_loss"], st["c_loss"]))
        agent.vec_env.env.hip.close()
    for other in results[1:]:
        for a, b in zip(results[0], other):
            assert torch.equal(a, b) if torch.is_tensor(a) else (a == b).all() if hasattr(a, "all") else a == b


def _dp_gpu_worker(rank, world, port, q):
    try:
        _dp_gpu_worker_body(rank, world, port, q)
    except BaseException:          # report instead of leaving the parent to wait for its timeout
        import traceback
        q.put((rank, "error", traceback.format_exc()))
        raise


def _dp_gpu_worker_body(rank, world, port, q):
    import os
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent

    class Args:
        envs = 1024; minibatches = 4; graph = 1; task = "hovering"; ctl = "rate"; tuned_gemms = 1
    params = bench.build_params(Args, world)
    c = params["config"]
    c["dist_backend"] = "gloo"            # both ranks share the one GPU of the test box; the collective goes through gloo
    c["device"] = "cuda:0"
    torch.manual_seed(100 + rank)         # different initialisation per rank: broadcast_parameters must repair it
    agent = A2CAgent("dp", params)
    agent.init_tensors()
    assert agent.multi_gpu and agent._fused_step is not None and agent._fused_rollout is not None
    # 6 144-sample minibatches: the launch-bound regime - minibatch hipGraphs, split at the gradient all-reduce under multi_gpu
    assert agent.env_config["env_id_offset"] == rank * 1024 and agent._graph_update
    agent.obs = agent.env_reset()
    agent.broadcast_parameters()
    for ep in range(1, 4):
        agent.epoch_num = ep
        st = agent.train_epoch()
    # epochs 2 and 3 replayed graph A (forward / backward / reductions) + eager all-reduce + graph B (average, clip, Adam) for the
    # statistics-off minibatches; the first mini-epoch's (normaliser moments all-reduced inside the forward) stayed eager
    assert "tail" in agent._upd_graphs and any(k != "tail" and k[1] is False for k in agent._upd_graphs)
    assert not any(k != "tail" and k[1] is True for k in agent._upd_graphs)
    q.put((rank, agent.flat_param.cpu().numpy().copy(), float(agent.optimizer.lr.item()), st["kl"],
           agent.model.running_mean_std.running_mean.cpu().numpy().copy(), agent.actions_buf[0, :4].cpu().numpy().copy()))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_fused_path_two_ranks_one_gpu(lib):
    """The multi-GPU code path with the fused rollout / hand-scheduled update: two ranks (sharing this box's one GPU, gloo
    for the gradient all-reduce) stay bit-identical replicas while rolling out different env shards."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_gpu_worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in range(2):
            got.append(q.get(timeout=150))
            assert not isinstance(got[-1][1], str), got[-1][2]
    except BaseException:
        for p in procs:
            p.terminate()
        raise
    res = dict((r[0], r[1:]) for r in got)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (p0, lr0, kl0, m0, a0), (p1, lr1, kl1, m1, a1) = res[0], res[1]
    assert np.array_equal(p0, p1) and lr0 == lr1 and kl0 == kl1 and np.array_equal(m0, m1)
    assert np.isfinite(p0).all() and not np.array_equal(a0, a1)      # same policy, different env shards / noise streams
