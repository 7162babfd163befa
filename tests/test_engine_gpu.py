"""GPU parity tests proper: the HIP engine, called through the C ABI, against the committed
golden vectors and the CPU oracle on the same seeded inputs.

Tolerance (fp16 trunk with fp32 accumulation vs fp32/float64 reference), stated per
SURVEY.md §8c and tightened after measurement (random-init nets: logit sigma ~0.35, largest
move probability ~0.012; worst case measured on MI355X over all 13 fixtures, round 2,
tools/gpu_parity_stats.py: logits 4.9e-3 (b8c128nbt), 362/800-way probabilities 3.3e-5,
outcome probabilities 2.6e-4, KL 6.9e-7):
  * every raw head output (logits, ownership, q6_err, gamma): |d| <= LOGIT_TOL = 6e-3 abs
    (1.7 % of the logit sigma), or, for an output of large magnitude (score logits reach +-10 on a
    random-init net), |d| <= LOGIT_REL = 1e-3 of the reference value — two fp16 half-ulps: the trunk's
    activations are fp16, so its noise is relative; round 3's epilogue arithmetic, a different
    rounding pattern of the same precision, moved one score logit of 8.19 by 6.3e-3 = 7.7e-4 of it;
  * move / optimistic-move / score probabilities: |d| <= PROB_TOL = 5e-5 abs (0.4 % of the
    largest move probability); the two-way outcome distribution moves by up to a quarter of
    its logit-difference error: |d| <= VALUE_PROB_TOL = 5e-4;
  * KL(reference || engine) <= KL_TOL = 2e-6 for each of the four distributions;
  * the `_peaked` fixture (policy output layer scaled x12 -> max move probability up to 0.96,
    the regime of a trained net): policy logits scale with the output layer, so their bound
    is 12 x LOGIT_TOL (measured 3.0e-2); probabilities |d| <= 1e-2 (measured 3.3e-3),
    KL <= 2e-4 (measured 2.4e-5), argmax and top-3 identical.
The reference itself pins no network numerics ("parity unpinned").
"""
import os
import threading

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

LOGIT_TOL = 6e-3
LOGIT_REL = 1e-3
PROB_TOL = 5e-5
VALUE_PROB_TOL = 5e-4
KL_TOL = 2e-6
PEAK_PROB_TOL = 1e-2
PEAK_KL_TOL = 2e-4
# the last six are full-size BASELINE architectures: C1 b8c128nbt, C2 b12c128btl3, C3/C4
# b12c256btl3 (32 wide positions + a peaked-policy set), C5 b10c384nbt / b14c384btl3
NETS = ["test_b3c128btl2", "test_b3c128nbt", "test_b3c256btl1", "test_b3c256nbt", "test_b3c384btl3", "test_b3c384nbt",
        "test_b3c192classic", "test_b5c256nbt_i2", "test_b5c128btl1_i2", "test_b5c256btl2_i2", "test_b10c256btl1_i2", "b8c128nbt", "b12c128btl3", "b12c256btl3", "b12c256btl3_peaked", "b10c384nbt",
        "b14c384btl3"]
PROB_KEYS = ("move_probs", "value_probs", "score_probs", "opt_move_probs")


def _kl(p, q):
    p = np.asarray(p, np.float64)
    q = np.maximum(np.asarray(q, np.float64), 1e-300)
    m = p > 0
    return float((p[m] * np.log(p[m] / q[m])).sum())


def _logits_close(got, want, scale=1.0):
    """|d| <= max(LOGIT_TOL * scale, LOGIT_REL * |reference|), elementwise.  The relative clause is the larger one only
    where |reference| > LOGIT_TOL / LOGIT_REL = 6 (score logits; two fp16 half-ulps of the value) — everywhere else the
    absolute bound stands as it was.  Neither may be widened without a measured fp16-ulp argument."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return bool((np.abs(got - want) <= np.maximum(LOGIT_TOL * scale, LOGIT_REL * np.abs(want))).all())


def _check(raw, res, ref_raw, ref, peak=0.0):
    assert not np.isnan(raw).any()
    pol = peak if peak else 1.0            # the policy logits scale with the peaked output layer
    assert _logits_close(raw[:724], ref_raw[:724], pol)
    assert _logits_close(raw[724:1887], ref_raw[724:1887])
    assert abs(raw[1887] - ref_raw[1887]) <= LOGIT_TOL and abs(raw[1888] - ref_raw[1888]) <= LOGIT_TOL
    for key in PROB_KEYS:
        got = np.ctypeslib.as_array(getattr(res, key))
        policy = key in ("move_probs", "opt_move_probs")
        ptol, ktol = (PEAK_PROB_TOL, PEAK_KL_TOL) if (peak and policy) else (PROB_TOL, KL_TOL)
        if key == "value_probs":
            ptol = VALUE_PROB_TOL
        assert np.abs(got - ref[key]).max() <= ptol, key
        assert _kl(ref[key], got) <= ktol, key
        assert abs(got.sum() - 1.0) < 1e-4, key
    if peak:
        mp = np.ctypeslib.as_array(res.move_probs)
        assert mp.argmax() == ref["move_probs"].argmax()
        assert set(np.argsort(mp)[-3:]) == set(np.argsort(ref["move_probs"])[-3:])
    assert np.array_equal(np.ctypeslib.as_array(res.move_logits), raw[:362])


def _weights_for(weight_files, name):
    """.p3w of a golden fixture: `<config>_peaked` = the config with the policy output layer x12."""
    if name.endswith("_peaked"):
        return weight_files(name[:-len("_peaked")], peak=12.0)
    return weight_files(name)


@pytest.mark.parametrize("name", NETS)
def test_engine_matches_golden(built, weight_files, name):
    from p3achygo_amd import engine
    g, pos = load_golden(name)
    wpath = _weights_for(weight_files, name)
    peak = float(g["peak"])
    assert (peak > 0) == name.endswith("_peaked")
    eng = engine.create_engine(engine.kind_from_engine_path(wpath), wpath, max(8, len(pos)), 1)
    assert eng.kind() == engine.Kind.kHip and eng.path() == wpath
    for i in range(len(pos)):
        eng.LoadBatch(i, pos[i:i + 1])
    eng.RunInference()
    for i in range(len(pos)):
        ref = {k: g[k][i] for k in PROB_KEYS}
        _check(eng.get_raw(i), eng.GetBatch(i), g["raw"][i], ref, peak)
        own = eng.GetOwnership(i)
        assert np.abs(own - g["raw"][i][1526:1887]).max() <= LOGIT_TOL
    eng.close()


@pytest.mark.parametrize("name,batch", [("b12c128btl3", 256), ("b10c384nbt", 96), ("b14c384btl3", 96)])
def test_baseline_configs_full_size(built, weight_files, name, batch):
    """BASELINE configs C2 (b12c128btl3, 256 games, batch 256) and C5 (b10c384nbt / b14c384btl3)
    at full depth: the golden positions placed at scattered slots of a full batch of other
    positions must reproduce the float64 fixture, and a strided sample of the rest must agree
    with the CPU oracle."""
    from oracle import oracle
    from p3achygo_amd import engine, features
    g, gpos = load_golden(name)
    wpath = weight_files(name)
    fill = features.random_positions(batch, seed=41, n_games=16, max_moves=300, komis=(7.5, -7.5, 0.5))
    slots = [int(s) for s in np.linspace(0, batch - 1, len(gpos)).round()]
    pos = fill.copy()
    pos[slots] = gpos
    eng = engine.create_engine(engine.kind_from_engine_path(wpath), wpath, batch, 1)
    eng.load_all(pos)
    eng.RunInference()
    for k, s in enumerate(slots):
        ref = {key: g[key][k] for key in PROB_KEYS}
        _check(eng.get_raw(s), eng.GetBatch(s), g["raw"][k], ref)
    sample = [s for s in range(3, batch, batch // 6) if s not in slots][:6]
    res, raw = oracle.OracleNet(wpath).forward_features(pos[sample], nthreads=8)
    for k, s in enumerate(sample):
        ref = {key: np.ctypeslib.as_array(getattr(res[k], key)) for key in PROB_KEYS}
        _check(eng.get_raw(s), eng.GetBatch(s), raw[k], ref)
    eng.close()


@pytest.mark.parametrize("name,n", [("test_b3c256btl1", 37), ("test_b3c128nbt", 41)])
def test_engine_matches_oracle_ragged_batch(built, weight_files, name, n):
    """Odd batch sizes (ragged last workgroup, 2-positions-per-workgroup tail) vs the oracle."""
    from oracle import oracle
    from p3achygo_amd import engine, features
    pos = features.random_positions(n, seed=5, n_games=8)
    eng = engine.HipEngine(weight_files(name), 64)
    eng.load_all(pos)
    eng.RunInference()
    res, raw = oracle.OracleNet(weight_files(name)).forward_features(pos, nthreads=8)
    for i in range(n):
        got = eng.get_raw(i)
        assert _logits_close(got[:1887], raw[i][:1887]), i
        mp = np.ctypeslib.as_array(eng.GetBatch(i).move_probs)
        assert np.abs(mp - np.ctypeslib.as_array(res[i].move_probs)).max() <= PROB_TOL
    eng.close()


def test_slot_compaction_and_reuse(built, weight_files):
    """Only loaded slots are evaluated; results land in the slot that was loaded, across
    several runs with different subsets (the contract nn_interface_sync_test.cc:78-172
    checks with its CountingEngine: no stale result, no slot mix-up)."""
    from oracle import oracle
    from p3achygo_amd import engine, features
    name = "test_b3c128btl2"
    pos = features.random_positions(12, seed=9, n_games=6)
    _, raw = oracle.OracleNet(weight_files(name)).forward_features(pos, nthreads=8)
    eng = engine.HipEngine(weight_files(name), 16)
    rng = np.random.default_rng(0)
    for rnd in range(4):
        slots = rng.choice(16, size=5, replace=False)
        which = rng.choice(12, size=5, replace=False)
        for s, w in zip(slots, which):
            eng.LoadBatch(int(s), pos[w:w + 1])
        eng.RunInference()
        for s, w in zip(slots, which):
            got = np.ctypeslib.as_array(eng.GetBatch(int(s)).move_logits)
            assert np.abs(got - raw[w][:362]).max() <= LOGIT_TOL
        others = [s for s in range(16) if s not in set(int(x) for x in slots)]
        with pytest.raises(engine.EngineError):
            eng.GetBatch(others[0])
    eng.close()


def test_concurrent_load_and_get(built, weight_files):
    """LoadBatch/GetBatch from many threads, each on its own slot (threading contract of
    nn_interface.cc:276 / nn_interface.h:254)."""
    from oracle import oracle
    from p3achygo_amd import engine, features
    name = "test_b3c128btl2"
    n = 32
    pos = features.random_positions(n, seed=21, n_games=8)
    _, raw = oracle.OracleNet(weight_files(name)).forward_features(pos, nthreads=8)
    eng = engine.HipEngine(weight_files(name), n)
    ths = [threading.Thread(target=eng.LoadBatch, args=(i, pos[i:i + 1])) for i in range(n)]
    [t.start() for t in ths]; [t.join() for t in ths]
    eng.RunInference()
    out = [None] * n

    def get(i):
        out[i] = np.ctypeslib.as_array(eng.GetBatch(i).move_logits).copy()
    ths = [threading.Thread(target=get, args=(i,)) for i in range(n)]
    [t.start() for t in ths]; [t.join() for t in ths]
    for i in range(n):
        assert np.abs(out[i] - raw[i][:362]).max() <= LOGIT_TOL
    eng.close()


@pytest.mark.parametrize("name", ["b12c256btl3", "b12c128btl3", "b8c128nbt"])
def test_full_batch_properties(built, weight_files, name):
    """BASELINE size (1024 positions): size-independent properties — every probability
    vector sums to 1, results do not depend on batch position or batch size (a position
    evaluated alone equals the same position inside the full batch, bit for bit), and a
    strided sample agrees with the oracle.  At this size the C = 128 launches run two
    workgroups per CU taking turns at the higher wave priority (BlockArgs::pair_turns); the
    single-position run does not."""
    from oracle import oracle
    from p3achygo_amd import engine, features
    base = features.random_positions(64, seed=31, n_games=16)
    pos = np.tile(base, 16)
    eng = engine.HipEngine(weight_files(name, randomize=False), 1024)
    eng.load_all(pos)
    eng.RunInference()
    logits = np.stack([np.ctypeslib.as_array(eng.GetBatch(i).move_logits).copy() for i in range(1024)])
    probs = np.stack([np.ctypeslib.as_array(eng.GetBatch(i).move_probs).copy() for i in range(0, 1024, 7)])
    assert not np.isnan(logits).any()
    assert np.abs(probs.sum(1) - 1).max() < 1e-4
    for rep in range(1, 16):
        assert np.array_equal(logits[:64], logits[64 * rep:64 * (rep + 1)])
    eng.LoadBatch(5, base[17:18])
    eng.RunInference()
    assert np.array_equal(np.ctypeslib.as_array(eng.GetBatch(5).move_logits), logits[17])
    idx = [0, 21, 42, 63]
    _, raw = oracle.OracleNet(weight_files(name, randomize=False)).forward_features(base[idx], nthreads=4)
    for k, i in enumerate(idx):
        assert np.abs(logits[i] - raw[k][:362]).max() <= LOGIT_TOL
    eng.close()


def test_selfplay_host_on_hip_engine(built, weight_files):
    """The self-play host (libp3host.so) binds the engine through the C ABI with dlopen and
    keeps two engines busy; every loaded slot is evaluated exactly once per batch."""
    from p3achygo_amd import host_api
    st = host_api.selfplay_run(weight_files("test_b3c128btl2"), num_games=128, num_threads=4, seconds=1.5,
                               default_n=8, default_k=4, selected_n=8, selected_k=4, max_moves=60,
                               warmup_batches=2, seed=5)
    assert st.positions > 1000 and st.moves > 0 and st.games > 0
    assert abs(st.positions - st.batches * 64) <= 128
    mv, bs, ws, ev = host_api.selfplay_one_game(weight_files("test_b3c128btl2"), 8, 4, 80, seed=9)
    mv2, *_ = host_api.selfplay_one_game(weight_files("test_b3c128btl2"), 8, 4, 80, seed=9)
    assert len(mv) > 10 and np.array_equal(mv, mv2)   # deterministic on the GPU too


def test_two_lanes_of_one_group_play_the_same_games_on_hip_engines(built, weight_files):
    """BASELINE configs[2] as stated, scaled down: ONE group of games over two engine instances filled in turn
    (host_api.set_lanes(2, 4), up to four playouts of a search waiting for results at once) plays, game runner by game
    runner, the same first game as the same group over one engine with one evaluation in flight — on the GPU, where a
    position's result must not depend on which batch, row or engine instance evaluated it — and keeps its batches full."""
    from p3achygo_amd import host_api
    host_api.set_policy(init_state_sampling=False)
    host_api.set_groups(1)

    def run(lanes, depth, steps):
        host_api.set_lanes(lanes, depth)
        host_api.set_step_limit(steps)
        st = host_api.selfplay_run(weight_files("test_b3c128btl2"), 96, 4, 0.0, default_n=16, default_k=4, selected_n=16,
                                   selected_k=4, max_moves=24, warmup_batches=1, seed=5)
        return host_api.last_first_game_digests(), st.positions / st.batches / 96

    try:
        base, fill1 = run(1, 1, 600)
        got, fill2 = run(2, 4, 720)
    finally:
        host_api.set_lanes(1, 1)
        host_api.set_step_limit(0)
        host_api.set_groups(2)
        host_api.set_policy()
    both = (got != 0) & (base != 0)
    assert both.sum() >= 88 and (got[both] == base[both]).all()
    assert fill1 == 1.0 and fill2 >= 0.88


def test_config_c1_game_with_bias_cache_on_hip_engine(built, weight_files):
    """BASELINE configs[0] on the GPU engine: v4's net (b8c128nbt) and search settings (Gumbel n=8
    k=4, bias_cache_lambda 0.3 / alpha 0.8, config/v4.json), one self-play game on one thread: it
    plays to the move limit, is reproducible, and the bias cache adjusts its roots."""
    from p3achygo_amd import host_api
    w = weight_files("b8c128nbt")
    try:
        host_api.set_bias_cache(0.3, 0.8)
        mv, bs, ws, ev = host_api.selfplay_one_game(w, 8, 4, 80, seed=4)
        pruned, adj = host_api.last_bias_counters()
        mv2, *_ = host_api.selfplay_one_game(w, 8, 4, 80, seed=4)
    finally:
        host_api.set_bias_cache(0.0, 0.8)
    assert len(mv) == 80 and np.array_equal(mv, mv2) and ev > 200
    assert pruned > 100 and adj > 0


_FUSE_CHILD = r"""
import sys, hashlib, tempfile, os
sys.path.insert(0, %r)
import numpy as np
from p3achygo_amd import engine, features, netspec
for name, batch in (("b12c256btl3", 300), ("b8c128nbt", 70), ("b12c256btl3", 5)):
    cfg = netspec.CONFIGS[name]
    path = os.path.join(tempfile.mkdtemp(), "n.p3w")
    netspec.save_p3w(path, cfg, netspec.generate_weights(cfg, randomize=True))
    pos = features.random_positions(batch, seed=3, n_games=7)
    eng = engine.HipEngine(path, batch)
    eng.load_all(pos); eng.RunInference()
    raw = np.stack([eng.get_raw(i) for i in range(batch)])
    print(name, batch, hashlib.sha256(raw.tobytes()).hexdigest())
    eng.close()
"""


@pytest.mark.gpu
def test_fused_block_launches_equal_one_launch_per_block(built):
    """One k_block launch runs up to six consecutive residual blocks (workgroups own their
    positions, so there is no grid-wide dependency between blocks).  Same arithmetic, same
    order: the outputs must be bit-identical to one launch per block (P3HIP_NO_FUSE), with
    several positions per workgroup (300), with two per workgroup slot (C=128, 70) and with
    fewer positions than workgroups (5)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for no_fuse in (False, True):
        env = dict(os.environ)
        env.pop("P3HIP_NO_FUSE", None)
        if no_fuse:
            env["P3HIP_NO_FUSE"] = "1"
        r = subprocess.run([sys.executable, "-c", _FUSE_CHILD % root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] and outs[0].count("\n") == 3


_BFUSE_CHILD = r"""
import sys, tempfile, os
sys.path.insert(0, %r)
import numpy as np
from p3achygo_amd import engine, features, netspec
out = {}
for name, batch in (("b12c256btl3", 300), ("test_b5c256nbt_i2", 5), ("test_b5c256btl2_i2", 37), ("test_b10c256btl1_i2", 11), ("test_b10c256btl1_i2", 300), ("b12c128btl3", 70), ("b8c128nbt", 600)):
    cfg = netspec.CONFIGS[name]
    path = os.path.join(tempfile.mkdtemp(), "n.p3w")
    netspec.save_p3w(path, cfg, netspec.generate_weights(cfg, randomize=True))
    pos = features.random_positions(batch, seed=5, n_games=9)
    eng = engine.HipEngine(path, batch)
    eng.load_all(pos); eng.RunInference()
    out[name + (":%%d" %% batch if name in out else "")] = np.stack([eng.get_raw(i) for i in range(batch)])
    eng.close()
np.savez(sys.argv[1], **out)
"""


@pytest.mark.gpu
def test_broadcast_convs_inside_block_launches_match_their_own_launches(built, tmp_path):
    """The broadcast blocks' conv_first / conv_last ride at the tail / head of the neighbouring
    k_block launches (kernels.hip, BC form).  Same weights and operand precision as the stand-alone
    k_conv1x1 launches (P3HIP_NO_BFUSE), different MFMA shape and summation order, so fp16 roundings of
    the intermediates fall differently: the two outputs (each within LOGIT_TOL of the float64 oracle in
    the parity tests above) stay within LOGIT_TOL of each other — an indexing or weight-order slip
    would show as O(1) — for one and several positions per workgroup, btl and nbt blocks, both C = 128
    workgroup forms.  Likewise the C = 256 launches' fused dense against k_bdense as its own launch."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for label, extra in (("fused", {}), ("own", {"P3HIP_NO_BFUSE": "1"}), ("fused_wg8", {"P3HIP_C128_WG8": "1"}),
                         ("dense_own", {"P3HIP_NO_DFUSE": "1"}), ("not_joined", {"P3HIP_NO_JOIN": "1"})):
        env = dict(os.environ)
        for k in ("P3HIP_NO_BFUSE", "P3HIP_C128_WG8", "P3HIP_NO_FUSE", "P3HIP_NO_DFUSE", "P3HIP_NO_JOIN"):
            env.pop(k, None)
        env.update(extra)
        path = str(tmp_path / (label + ".npz"))
        r = subprocess.run([sys.executable, "-c", _BFUSE_CHILD % root, path], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[label] = np.load(path)
    for name in res["own"].files:
        a, b, c = res["fused"][name], res["own"][name], res["fused_wg8"][name]
        assert not np.isnan(a).any() and not np.isnan(c).any()
        assert _logits_close(a[:, :1889], b[:, :1889]), name
        assert _logits_close(c[:, :1889], b[:, :1889]), name
        # C = 256: the broadcast dense rides in the tail as well (tail_dense; P3HIP_NO_DFUSE = k_bdense as its
        # own launch, t through HBM): conv_first taken transposed, the dense's K over the padded board rows
        d = res["dense_own"][name]
        assert not np.isnan(d).any()
        assert _logits_close(a[:, :1889], d[:, :1889]), name
        # ... and with everything of the broadcast blocks fused the whole trunk is ONE launch (joined runs: a position
        # goes through all its blocks in one workgroup; P3HIP_NO_JOIN = one launch per run): the same code in the
        # same order on the same values
        assert np.array_equal(a, res["not_joined"][name]), name


_DIRECT_CHILD = r"""
import sys, os, tempfile
sys.path.insert(0, %r)
import numpy as np
from p3achygo_amd import engine, features, netspec
cfg = netspec.CONFIGS["test_b3c256btl1"]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg, randomize=True))
pos = features.random_positions(70, seed=41, n_games=10)
out = []
for flags in (0, engine.FLAG_LAUNCH_GRAPH):
    eng = engine.HipEngine(path, 64, flags=flags)
    for rnd in range(4):
        idx = list(range(64)) if rnd != 2 else [5, 9, 33]          # full batches (rows = slots) and a ragged, compacted one
        for i in idx:
            eng.LoadBatch(i, pos[(i + rnd) %% 70:(i + rnd) %% 70 + 1])
        eng.RunInference()
        for i in idx:
            r = eng.GetBatch(i)
            out.append(np.concatenate([np.ctypeslib.as_array(getattr(r, k)).ravel() for k in
                                       ("move_logits", "move_probs", "value_probs", "score_probs", "opt_move_probs")] + [[r.err2_outcome]]))
    eng.close()
np.save(sys.argv[1], np.stack(out))
"""


@pytest.mark.gpu
def test_result_records_written_by_the_heads_kernel_equal_the_copied_ones(built, tmp_path):
    """p3hip_run's result records reach the pinned host buffer from inside the heads kernel (HeadsArgs::res; the D2H copy of
    trt_engine.cc:283-297 folded into the kernel that produces the values).  Bit for bit what the strided copy behind the
    forward pass delivers (P3HIP_NO_DIRECT_RESULTS=1), for full batches, a compacted ragged run, and under the launch graph."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for extra in ({}, {"P3HIP_NO_DIRECT_RESULTS": "1"}):
        env = dict(os.environ)
        env.pop("P3HIP_NO_DIRECT_RESULTS", None)
        env.update(extra)
        path = str(tmp_path / ("d%d.npy" % len(res)))
        r = subprocess.run([sys.executable, "-c", _DIRECT_CHILD % root, path], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(path))
    assert res[0].shape == res[1].shape and res[0].shape[0] == 2 * (3 * 64 + 3)
    assert not np.isnan(res[0]).any() and np.array_equal(res[0], res[1])


_BLOCKW_CHILD = r"""
import sys, os
sys.path.insert(0, %r)
import numpy as np
from conftest import load_golden
from p3achygo_amd import engine, features, netspec
out = {}
for name in ("test_b3c256btl1", "test_b5c256btl2_i2", "test_b10c256btl1_i2", "b12c256btl3"):
    cfg = netspec.CONFIGS[name]
    gold, pos = load_golden(name)
    W = netspec.generate_weights(cfg, randomize=True)
    path = os.path.join(sys.argv[1], name + ".p3w")
    netspec.save_p3w(path, cfg, W)
    pad = features.random_positions(300 - len(pos), seed=9, n_games=11)
    allpos = np.concatenate([pos, pad])
    eng = engine.HipEngine(path, len(allpos))
    eng.load_all(allpos); eng.RunInference()
    out[name] = np.stack([eng.get_raw(i) for i in range(len(allpos))])
    eng.close()
np.savez(sys.argv[2], **out)
"""


@pytest.mark.gpu
def test_hand_scheduled_block_kernel_matches_the_fixtures_and_the_hip_kernels(built, tmp_path):
    """k_blockw (csrc/asm/blockw_gen.py, opt-in: P3HIP_BLOCKW=1): the runs of C = 256 btl blocks by the generated
    one-wave-per-SIMD assembly kernel, the broadcast blocks as their own launches.  Its BN scales ride in the fp16 weights
    (one rounding of w * scale instead of rounding w and multiplying in fp32), so it differs from the HIP kernels by fp16
    roundings of the intermediates — both stay inside the logit bound against the float64 fixtures, and inside it of each
    other over 300 positions (4 positions per workgroup slot at 256 CUs is not reached; more than one, with 300)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for label, extra in (("hip", {"P3HIP_NO_BFUSE": "1"}), ("blockw", {"P3HIP_BLOCKW": "1"})):
        env = dict(os.environ)
        for k in ("P3HIP_NO_BFUSE", "P3HIP_BLOCKW", "P3HIP_BLOCKW_DIAG"):
            env.pop(k, None)
        env.update(extra)
        env["PYTHONPATH"] = os.path.join(root, "tests") + os.pathsep + env.get("PYTHONPATH", "")
        path = str(tmp_path / (label + ".npz"))
        r = subprocess.run([sys.executable, "-c", _BLOCKW_CHILD % root, str(tmp_path), path], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[label] = np.load(path)
    for name in res["hip"].files:
        a, b = res["blockw"][name], res["hip"][name]
        assert not np.isnan(a).any()
        assert _logits_close(a[:, :1889], b[:, :1889]), name
        gold, gpos = load_golden(name)
        n = len(gpos)
        assert _logits_close(a[:n, :1887], np.asarray(gold["raw"], np.float64)[:, :1887]), name


@pytest.mark.gpu
def test_thread_per_game_nn_interface_on_hip_engine(built, weight_files):
    """The reference's blocking NNInterface (host/nn_interface.h) over the HIP engine: 32 game
    threads batched by the infer thread get bit-identical results to the same games run one
    after another (same symmetry draws), i.e. slots never mix and batching changes nothing."""
    import ctypes as C
    from p3achygo_amd import engine, host_api
    L = host_api.lib()
    L.p3host_nn_new.restype = C.c_void_p
    L.p3host_nn_new.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_long, C.c_long, C.c_int,
                                C.c_char_p, C.c_int]
    L.p3host_nn_free.argtypes = [C.c_void_p]
    L.p3host_nn_num_inferences.restype = C.c_long
    L.p3host_nn_num_inferences.argtypes = [C.c_void_p]
    L.p3host_nn_play_threads_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_void_p]
    T, M = 32, 10
    outs, infs = [], []
    for sequential in (0, 1):
        err = C.create_string_buffer(512)
        nn = L.p3host_nn_new(1, engine.LIB_PATH.encode(), weight_files("test_b3c128btl2").encode(), 0, T, 400, 0, 1,
                             err, 512)
        assert nn, err.value.decode()
        out = (engine.Result * (T * M))()
        L.p3host_nn_play_threads_ex(nn, T, M, 40, sequential, C.byref(out))
        infs.append(L.p3host_nn_num_inferences(nn))
        L.p3host_nn_free(nn)
        words = np.frombuffer(out, dtype=np.uint32).reshape(T * M, -1).copy()
        assert words.shape[1] == 1892
        words[:, [1526, 1527, 1891]] = 0        # struct padding (alignas(16) opt_move_probs, tail)
        outs.append(words)
    assert np.array_equal(outs[0], outs[1])
    assert infs[0] < infs[1] == T * M          # threaded run batched the leaves
    probs = outs[0].view(np.float32)[:, 362:724]
    assert np.allclose(probs.sum(1), 1.0, atol=1e-3)


@pytest.mark.gpu
def test_eval_match_on_hip_engines(built, weight_files):
    """Config 5 plumbing: two different networks, each behind its own HIP
    engine instance, play evaluation games with the batch parallel search."""
    from p3achygo_amd import host_api
    st = host_api.eval_match(weight_files("test_b3c384nbt"), weight_files("test_b3c256btl1"), num_games=8,
                             visits_per_move=16, leaves_per_round=4, max_moves=24, num_threads=4, seed=2)
    assert st.games == 8 and st.cur_wins + st.cand_wins + st.draws == 8
    assert st.visits >= 16 * (st.moves - 8) and st.positions > 0 and st.batches > 0


@pytest.mark.gpu
def test_thread_per_game_match_with_the_nn_cache_in_hbm(built, weight_files):
    """The reference's shape of an evaluation match (one thread per game, two NNInterfaces, threaded search) with
    the NN cache in each engine's HBM table instead of the interfaces' host LRUs (host_api.set_device_nn_cache):
    the match completes, every game has a result, every evaluation went through the keyed path, and positions met
    again (transpositions inside a search; how many depends on the threads' timing) are served from the tables."""
    from p3achygo_amd import host_api
    host_api.set_device_nn_cache(14)
    try:
        weights = weight_files
        st = host_api.eval_match_threads(weights("test_b3c128btl2"), weights("test_b3c128nbt"), num_games=6,
                                         visits_per_move=24, threads_per_game=4, max_moves=16, cache_size=0, seed=5)
    finally:
        host_api.set_device_nn_cache(0)
    assert st.games == 6 and st.cur_wins + st.cand_wins + st.draws == 6 and st.moves > 6
    lookups, hits = host_api.device_nn_cache_lookups(), host_api.device_nn_cache_hits()
    assert lookups > 100 and 0 <= hits < lookups
    # the batching scheduler's match puts the same tables behind its per-game host caches
    host_api.set_device_nn_cache(12)
    try:
        st = host_api.eval_match(weights("test_b3c128btl2"), weights("test_b3c128nbt"), num_games=6, visits_per_move=16,
                                 leaves_per_round=4, max_moves=16, num_threads=4, seed=5)
    finally:
        host_api.set_device_nn_cache(0)
    assert st.games == 6 and st.cur_wins + st.cand_wins + st.draws == 6
    assert host_api.device_nn_cache_lookups() == st.positions and 0 <= host_api.device_nn_cache_hits() < st.positions


@pytest.mark.gpu
def test_edge_cases_empty_single_and_flags(built, weight_files):
    """Empty run (no slot loaded) is a no-op, a batch of one works, RUN_ALL_SLOTS evaluates the
    static batch like the TRT engine (trt_engine.cc:238-304), unsupported architectures and
    versions fail at creation with a message, and ownership is served on request."""
    from oracle import oracle
    from p3achygo_amd import engine, features, netspec
    name = "test_b3c256btl1"
    pos = features.random_positions(3, seed=4, n_games=3)
    _, raw = oracle.OracleNet(weight_files(name)).forward_features(pos, nthreads=4)
    eng = engine.HipEngine(weight_files(name), 1)
    eng.RunInference()                                   # nothing loaded
    with pytest.raises(engine.EngineError):
        eng.GetBatch(0)
    eng.LoadBatch(0, pos[0:1])
    eng.RunInference()
    assert np.abs(np.ctypeslib.as_array(eng.GetBatch(0).move_logits) - raw[0][:362]).max() <= LOGIT_TOL
    own = eng.GetOwnership(0)
    assert np.abs(np.asarray(own) - np.tanh(raw[0][1526:1887])).max() <= 2e-2 or \
        np.abs(np.asarray(own) - raw[0][1526:1887]).max() <= 2e-2
    with pytest.raises(engine.EngineError):
        eng.LoadBatch(1, pos[0:1])                       # slot out of range
    eng.close()
    eng = engine.HipEngine(weight_files(name), 4, flags=engine.FLAG_RUN_ALL_SLOTS)
    eng.LoadBatch(2, pos[1:2])
    eng.RunInference()
    assert np.abs(np.ctypeslib.as_array(eng.GetBatch(2).move_logits) - raw[1][:362]).max() <= LOGIT_TOL
    eng.GetBatch(0)                                      # every slot of the static batch was run
    eng.close()
    with pytest.raises(engine.EngineError):
        engine.HipEngine(weight_files(name), 4, version=0)
    import os
    import tempfile
    cfg = netspec.CONFIGS["tiny"]                        # C = 16: not a HIP-engine architecture
    p = os.path.join(tempfile.mkdtemp(), "tiny.p3w")
    netspec.save_p3w(p, cfg, netspec.generate_weights(cfg))
    with pytest.raises(engine.EngineError, match="unsupported architecture"):
        engine.HipEngine(p, 4)


@pytest.mark.gpu
def test_repeated_and_concurrent_runs_are_bit_identical(built, weight_files):
    """Race screen for the hand-synchronised kernels (counted vmcnt / lgkmcnt, raw barriers): the
    same batch gives bit-identical raw outputs on every run, also while a second engine instance
    runs on the same device from another thread (how the self-play host drives it)."""
    import hashlib
    from p3achygo_amd import engine, features
    name = "b12c256btl3"
    batch = 512
    pos = np.tile(features.random_positions(64, seed=2, n_games=16), batch // 64).copy()

    def digest(eng):
        h = hashlib.sha1()
        for s in (0, 1, batch // 2, batch - 1, 77):
            h.update(eng.get_raw(s).tobytes())
        return h.hexdigest()

    def worker(out, iters):
        eng = engine.HipEngine(weight_files(name), batch)
        ds = set()
        for _ in range(iters):
            eng.load_all(pos)
            eng.RunInference()
            ds.add(digest(eng))
        eng.close()
        out.append(ds)

    solo = []
    worker(solo, 20)
    assert len(solo[0]) == 1
    outs = []
    ths = [threading.Thread(target=worker, args=(outs, 20)) for _ in range(2)]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert len(outs) == 2 and outs[0] == outs[1] == solo[0]


@pytest.mark.gpu
def test_first_run_of_an_engine_created_while_the_device_is_busy(built, weight_files):
    """p3hip_create uploads the weight arena and clears its buffers through the engine's own (non-blocking) stream and
    synchronises it before it returns.  It used to leave both on the null stream, which that stream never waits for:
    an engine created while another kept the GPU busy could evaluate its first batch on weights that had not landed
    (whole result rows off by O(1), seen in 4 of 6 runs of the test above once the block weight streams had moved to the
    end of the arena).  Here: engines created one after another while a second thread runs forward passes back to back;
    the FIRST batch of each must equal the reference bit for bit."""
    from p3achygo_amd import engine, features
    name = "b12c256btl3"
    batch = 256
    pos = features.random_positions(batch, seed=21, n_games=16)
    ref_eng = engine.HipEngine(weight_files(name), batch)
    ref_eng.load_all(pos); ref_eng.RunInference()
    ref = np.stack([ref_eng.get_raw(s) for s in (0, 1, 100, batch - 1)])
    stop = threading.Event()

    def busy():
        ref_eng.upload()
        while not stop.is_set():
            for _ in range(8):
                ref_eng.forward_resident(batch)
            ref_eng.sync()

    th = threading.Thread(target=busy)
    th.start()
    try:
        for _ in range(6):
            eng = engine.HipEngine(weight_files(name), batch)
            eng.load_all(pos); eng.RunInference()
            got = np.stack([eng.get_raw(s) for s in (0, 1, 100, batch - 1)])
            eng.close()
            assert np.array_equal(got, ref)
    finally:
        stop.set()
        th.join()
        ref_eng.close()


@pytest.mark.gpu
def test_device_nn_cache_serves_hits_bit_identical_and_evaluates_only_misses(built, weight_files):
    """On-device NN cache (p3hip_cache_*; the reference's per-thread LRU of cc/nn/nn_interface.cc:107-132 moved into
    HBM).  A keyed position is evaluated once; later runs return the stored record bit for bit, with the symmetry
    it was stored under, from whatever slot asks; unkeyed slots are never cached; runs that mix hits, misses,
    duplicate new keys and unkeyed slots equal an engine without the cache; a table far smaller than the key set
    (forced evictions) never returns another key's record."""
    from p3achygo_amd import engine, features
    name = "test_b3c128btl2"
    path = weight_files(name, randomize=True)
    pos = features.random_positions(96, seed=77, n_games=24)
    ref = engine.HipEngine(path, 96)
    ref.load_all(pos)
    ref.LoadBatchKeyed(5, pos[5:6], 11, 22, symmetry=6)    # no table: a keyed load is a plain load that remembers its symmetry
    ref.RunInference()
    want = np.stack([ref.get_raw(i) for i in range(96)])
    _, sym, hit = ref.GetBatchKeyed(5)
    assert sym == 6 and not hit
    _, sym, hit = ref.GetBatchKeyed(6)
    assert sym == 0 and not hit
    ref.close()
    key = lambda i: (0x9E3779B97F4A7C15 * (i + 1) & (2**64 - 1), 0xC2B2AE3D27D4EB4F * (i + 7) & (2**64 - 1))

    eng = engine.HipEngine(path, 64)
    eng.EnableCache(10)
    # run 1: 48 keyed positions (symmetry = i % 8 recorded with each) + 16 unkeyed: everything is evaluated
    for s in range(48):
        eng.LoadBatchKeyed(s, pos[s:s + 1], *key(s), symmetry=s % 8)
    for s in range(48, 64):
        eng.LoadBatch(s, pos[s:s + 1])
    eng.RunInference()
    for s in range(64):
        raw = eng.get_raw(s)
        _, sym, hit = eng.GetBatchKeyed(s)
        assert np.array_equal(raw, want[s]) and not hit and sym == (s % 8 if s < 48 else 0)
    assert eng.cache_stats() == {"lookups": 48, "hits": 0, "stored": 48, "entries": 1024}
    # run 2: the same keys from other slots (features deliberately those of ANOTHER position: a hit never looks at
    # them), 8 new keys of which two are the same key twice, 8 unkeyed
    for s in range(48):
        k = 47 - s
        eng.LoadBatchKeyed(s, pos[(k + 1) % 96:(k + 1) % 96 + 1], *key(k), symmetry=7)
    for s in range(48, 56):
        p = 64 + min(s - 48, 6)            # slots 54 and 55 both carry position / key 70
        eng.LoadBatchKeyed(s, pos[p:p + 1], *key(p), symmetry=3)
    for s in range(56, 64):
        eng.LoadBatch(s, pos[s:s + 1])
    eng.RunInference()
    for s in range(48):
        k = 47 - s
        raw = eng.get_raw(s)
        res, sym, hit = eng.GetBatchKeyed(s)
        assert hit and sym == k % 8 and np.array_equal(raw, want[k])
        assert np.array_equal(np.ctypeslib.as_array(res.move_logits), want[k][:362])
    for s in range(48, 56):
        p = 64 + min(s - 48, 6)
        _, sym, hit = eng.GetBatchKeyed(s)
        assert not hit and sym == 3 and np.array_equal(eng.get_raw(s), want[p])
    for s in range(56, 64):
        assert np.array_equal(eng.get_raw(s), want[s])
        eng.GetBatch(s)
    st = eng.cache_stats()
    assert st["lookups"] == 48 + 56 and st["hits"] == 48 and st["stored"] == 48 + 7
    # run 3: a partial run (ragged: 5 slots), all hits including the duplicated key
    for s, p in enumerate([70, 64, 3, 69, 40]):
        eng.LoadBatchKeyed(s, pos[0:1], *key(p), symmetry=0)
    eng.RunInference()
    for s, p in enumerate([70, 64, 3, 69, 40]):
        _, sym, hit = eng.GetBatchKeyed(s)
        assert hit and sym == (3 if p >= 64 else p % 8) and np.array_equal(eng.get_raw(s), want[p])
    eng.close()

    # 16 entries for 96 keys: whatever is evicted is simply evaluated again; a record is never served for the wrong key
    small = engine.HipEngine(path, 32)
    small.EnableCache(4)
    rng = np.random.default_rng(5)
    hits = 0
    for it in range(12):
        ids = rng.choice(96, 32, replace=False)
        for s, p in enumerate(ids):
            small.LoadBatchKeyed(s, pos[p:p + 1], *key(int(p)), symmetry=int(p) % 8)
        small.RunInference()
        for s, p in enumerate(ids):
            raw = small.get_raw(s)
            _, sym, hit = small.GetBatchKeyed(s)
            hits += hit
            assert sym == int(p) % 8 and np.array_equal(raw, want[p])
    assert 0 < hits < 12 * 32 and small.cache_stats()["hits"] == hits
    small.close()


def test_device_nn_cache_of_an_impossible_size_fails_cleanly(built, weight_files):
    """p3hip_cache_enable with a table larger than the GPU's memory (2^26 entries x 13.7 KB = 917 GB of the 288)
    returns an error and keeps nothing of what it had allocated before the records failed; a smaller table can
    then be enabled on the same engine and serves hits (round-2 advisor: the failed call used to leave its buffers
    behind and the second call leaked them)."""
    from p3achygo_amd import engine, features
    path = weight_files("test_b3c128btl2", randomize=True)
    pos = features.random_positions(8, seed=5, n_games=4)
    eng = engine.HipEngine(path, 8)
    with pytest.raises(engine.EngineError):
        eng.EnableCache(26)
    eng.EnableCache(10)
    for rnd in range(2):
        for i in range(8):
            eng.LoadBatchKeyed(i, pos[i:i + 1], 1000 + i, 7, symmetry=0)
        eng.RunInference()
        outs = [np.ctypeslib.as_array(eng.GetBatchKeyed(i)[0].move_logits).copy() for i in range(8)]
        if rnd == 0:
            first = outs
    assert all(np.array_equal(a, b) for a, b in zip(first, outs))
    st = eng.cache_stats()
    assert st["hits"] >= 8 and st["entries"] == 1 << 10     # the second round was served from the table
    eng.close()


def test_baseline_config_c5_eval_match_with_the_hbm_cache_on_the_full_size_nets(built, weight_files):
    """BASELINE configs[4] as a configuration: "v3-b10c384nbt / b14c384 large trunk, fp16, NN-eval cache on (cc/eval
    path)" — the reference's thread-per-game match (eval/main.cc:380-452) between the two FULL-DEPTH C = 384 nets with
    the NN cache in the engines' HBM tables behind the interfaces' LRUs.  Few games, few visits (a full-size forward
    pass of a handful of positions is latency-bound: this is a correctness test, the rate is tools/gpu_eval_threads_cache.py's).
      (i)  what the engines serve with the cache on is the nets' output: the golden positions of each net, loaded
           keyed into an engine with a table, first evaluation against the float64 fixture within the tolerances above;
      (ii) a hit is the first evaluation bit for bit (asked again under another symmetry and from other slots);
      (iii) the match itself completes through the keyed path, every game with a result, and lookups reach the tables."""
    from p3achygo_amd import engine, host_api
    nets = ("b10c384nbt", "b14c384btl3")
    for name in nets:
        g, pos = load_golden(name)
        eng = engine.create_engine(engine.Kind.kHip, weight_files(name), 8, 1)
        eng.EnableCache(12)
        n = len(pos)
        for i in range(n):
            eng.LoadBatchKeyed(i, pos[i:i + 1], 0x1000 + i, 0xabc, symmetry=i % 8)
        eng.RunInference()
        first = []
        for i in range(n):
            res, sym, hit = eng.GetBatchKeyed(i)
            assert not hit and sym == i % 8
            _check(eng.get_raw(i), res, g["raw"][i], {k: g[k][i] for k in PROB_KEYS})
            first.append(np.ctypeslib.as_array(res.move_logits).copy())
        for i in range(n):                                   # again: other slots, another symmetry asked for
            eng.LoadBatchKeyed(n + i, pos[i:i + 1], 0x1000 + i, 0xabc, symmetry=(i + 3) % 8)
        eng.RunInference()
        for i in range(n):
            res, sym, hit = eng.GetBatchKeyed(n + i)
            assert hit and sym == i % 8                      # the stored record, under the symmetry it was stored with
            assert np.array_equal(np.ctypeslib.as_array(res.move_logits), first[i])
        st = eng.cache_stats()
        assert st["hits"] == n and st["stored"] == n
        eng.close()
    host_api.set_device_nn_cache(12)
    try:
        st = host_api.eval_match_threads(weight_files(nets[0]), weight_files(nets[1]), num_games=4, visits_per_move=8,
                                         threads_per_game=3, max_moves=6, cache_size=1 << 12, seed=11)
    finally:
        host_api.set_device_nn_cache(0)
    assert st.games == 4 and st.cur_wins + st.cand_wins + st.draws == 4 and st.moves >= 4
    lookups, hits = host_api.device_nn_cache_lookups(), host_api.device_nn_cache_hits()
    assert lookups >= st.moves and 0 <= hits < lookups


def test_launch_graph_replays_the_full_batch_forward_bit_for_bit(built, weight_files):
    """P3HIP_FLAG_LAUNCH_GRAPH: a run over the full static batch replays one captured launch graph, as the
    reference's TensorRT engine does (trt_engine.cc:260-303: capture once, cudaGraphLaunch per RunInference); the
    first such run goes out kernel by kernel, the second is captured, later ones replay.  Same kernels: every run is
    bit-identical to an engine without the flag, runs over fewer slots in between (launched kernel by kernel) too."""
    from p3achygo_amd import engine, features
    path = weight_files("test_b3c256btl1", randomize=True)
    B = 64
    pos = features.random_positions(B, seed=21, n_games=16)
    ref = engine.HipEngine(path, B)
    gr = engine.HipEngine(path, B, flags=engine.FLAG_LAUNCH_GRAPH)

    def run(eng, idx):
        for i in idx:
            eng.LoadBatch(i, pos[i:i + 1])
        eng.RunInference()
        return [eng.get_raw(i).copy() for i in idx]

    full = list(range(B))
    want = run(ref, full)
    for rnd in range(5):                       # eager, capture, replay, replay, replay
        got = run(gr, full)
        assert all(np.array_equal(a, b) for a, b in zip(want, got)), rnd
        assert gr.graph_state() == (1 if rnd >= 1 else 0) and ref.graph_state() == 0
        if rnd == 2:                           # a ragged run between replays
            part = [3, 17, 40]
            assert all(np.array_equal(a, b) for a, b in zip(run(ref, part), run(gr, part)))
    ref.close()
    gr.close()


def test_launch_graph_with_the_nn_cache_never_replays_the_other_feature_buffer(built, weight_files):
    """The captured graph bakes k_init's feature pointer in, and a cached run points the engine at the cache's gathered
    copy of the misses around its forward pass (engine.cpp run_cached).  A graph captured there — a full batch of new keys —
    must not be replayed by p3hip_forward_resident / an uncached full run, which read the uploaded buffer, nor the other
    way round: the graph serves the buffer it was captured on, the other buffer goes out kernel by kernel."""
    from p3achygo_amd import engine, features
    path = weight_files("test_b3c256btl1", randomize=True)
    B = 32
    posA = features.random_positions(B, seed=31, n_games=8)
    posB = features.random_positions(B, seed=32, n_games=8)
    ref = engine.HipEngine(path, B)
    def plain(pos):
        ref.load_all(pos)
        ref.RunInference()
        return np.stack([ref.get_raw(i) for i in range(B)])
    wantA, wantB = plain(posA), plain(posB)
    ref.close()
    key = lambda r, i: ((0x9E3779B97F4A7C15 * (1000 * r + i + 1)) & (2**64 - 1), (0xC2B2AE3D27D4EB4F * (i + 7)) & (2**64 - 1))
    eng = engine.HipEngine(path, B, flags=engine.FLAG_LAUNCH_GRAPH)
    eng.EnableCache(12)
    for rnd in range(3):                       # three full batches of NEW keys: eager, capture (on the gathered copy), replay
        pos, want = (posA, wantA) if rnd % 2 == 0 else (posB, wantB)
        for i in range(B):
            eng.LoadBatchKeyed(i, pos[i:i + 1], *key(rnd, i), symmetry=0)
        eng.RunInference()
        assert all(np.array_equal(eng.get_raw(i), want[i]) for i in range(B)), rnd
    assert eng.graph_state() == 1
    # the uploaded buffer now holds posA (round 2); put posB there and run the forward pass over it directly
    eng.load_all(posB)
    eng.upload()
    eng.forward_resident(B)
    eng.sync()
    got = np.stack([eng.get_raw(i) for i in range(B)])
    assert np.array_equal(got, wantB)
    eng.close()


def test_bench_line_keeps_its_contract_and_its_self_checks(built):
    """bench.py's one JSON line on the GPU (a short run: 8 rounds, no CPU baseline, no counter passes): the driver's keys,
    a steady-state headline (per-batch time not below the trunk launch's own time, value not above what the forward pass
    alone does by more than the streams' overlap allows), the roofline object, and the chip's state beside the engine-only leg."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline",
                        "--no-pmc", "--no-extras", "--engine-steps", "60"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["unit"] == "positions/s" and d["dtype"] == "f16"
    assert "workload" in d["config"] and d["config"]["game_groups_per_gpu"] == 8
    assert d["engine_batches_completed"] >= 8 * 8 - 8 and d["mean_batch_fill"] > 0.99
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 2500.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0.2 < roof["frac"] < 0.6
    assert d["window_artifact"] is False and d["ms_per_batch"] >= 0.98 * roof["launch_ms"]
    assert d["value"] <= 1.06 * d["engine_only"]["value"]          # a self-play line cannot beat its own forward pass
    chip = d["engine_only"]["chip_during_loop"]
    if chip is not None:                                            # amdsmi present: clock, power and limiter residencies
        assert 500 < chip["gfx_clock_mhz_mean"] < 3000 and chip["socket_power_w_mean"] > 100
