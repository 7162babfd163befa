"""TF chunk recorder of the self-play host (p3achygo_amd/host/tf_recorder.h).

The cases follow the reference's cc/recorder/__tests__/tf_recorder_test.cc and
sel_mult_test.cc: same games (pass games / two-move games), same per-move records, same
assertions.  The chunk is read back with an independent reader written here: zlib stream,
TFRecord framing with masked CRC32C, and tf.Example parsed by google.protobuf from
descriptors built at run time (tensorflow itself is not installed)."""
import ctypes as C
import glob
import os
import struct
import zlib

import numpy as np
import pytest

from p3achygo_amd import host_api

NUM_MOVES, NUM_V_BUCKETS, PASS = 362, 51, 362   # move encoding: 1-based index, sign = colour


def crc32c_py(data: bytes) -> int:
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ 0x82F63B78 if crc & 1 else crc >> 1
    return crc ^ 0xFFFFFFFF


def masked(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def example_class():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="p3_example.proto", package="p3t", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def field(m, name, num, typ, label=T.LABEL_OPTIONAL, type_name=None, oneof=None):
        f = m.field.add(name=name, number=num, type=typ, label=label)
        if type_name:
            f.type_name = type_name
        if oneof is not None:
            f.oneof_index = oneof
        return f

    field(msg("BytesList"), "value", 1, T.TYPE_BYTES, T.LABEL_REPEATED)
    field(msg("FloatList"), "value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED)
    field(msg("Int64List"), "value", 1, T.TYPE_INT64, T.LABEL_REPEATED)
    feat = msg("Feature")
    feat.oneof_decl.add(name="kind")
    field(feat, "bytes_list", 1, T.TYPE_MESSAGE, type_name=".p3t.BytesList", oneof=0)
    field(feat, "float_list", 2, T.TYPE_MESSAGE, type_name=".p3t.FloatList", oneof=0)
    field(feat, "int64_list", 3, T.TYPE_MESSAGE, type_name=".p3t.Int64List", oneof=0)
    feats = msg("Features")
    entry = feats.nested_type.add(name="FeatureEntry")
    entry.options.map_entry = True
    field(entry, "key", 1, T.TYPE_STRING)
    field(entry, "value", 2, T.TYPE_MESSAGE, type_name=".p3t.Feature")
    field(feats, "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, ".p3t.Features.FeatureEntry")
    field(msg("Example"), "features", 1, T.TYPE_MESSAGE, type_name=".p3t.Features")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("p3t.Example"))


@pytest.fixture(scope="module")
def L(built):
    lib = host_api.lib()
    lib.p3host_tfrec_new.restype = C.c_void_p
    lib.p3host_tfrec_new.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
    lib.p3host_tfrec_free.argtypes = [C.c_void_p]
    lib.p3host_tfrec_flush.argtypes = [C.c_void_p]
    lib.p3host_tfrec_record.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 7
    lib.p3host_crc32c.restype = C.c_uint32
    lib.p3host_crc32c.argtypes = [C.c_char_p, C.c_size_t]
    return lib


class Recorder:
    def __init__(self, L, d, gen=0, worker="test"):
        self.L, self.dir = L, str(d)
        self.h = L.p3host_tfrec_new(self.dir.encode(), gen, worker.encode())

    def record(self, moves, pi=None, trainable=None, root_q=None, root_score=None, kld=None, dist=None,
               stats=None, komi=7.5):
        n = len(moves)

        def arr(x, dt, shape):
            if x is None:
                return None, None
            a = np.ascontiguousarray(np.asarray(x, dt).reshape(shape))
            return a, a.ctypes.data

        keep = []
        ptrs = []
        for x, dt, shape in ((pi, np.float32, (n, NUM_MOVES)), (trainable, np.uint8, (n,)),
                             (root_q, np.float32, (n,)), (root_score, np.float32, (n,)), (kld, np.float32, (n,)),
                             (dist, np.uint32, (n, NUM_V_BUCKETS)), (stats, np.float32, (n, 13))):
            a, p = arr(x, dt, shape)
            keep.append(a)
            ptrs.append(p)
        mv = np.asarray(moves, np.int32)
        assert self.L.p3host_tfrec_record(self.h, mv.ctypes.data, n, komi, *ptrs) == 0

    def flush(self):
        return self.L.p3host_tfrec_flush(self.h)

    def examples(self):
        Example = example_class()
        out = []
        for path in sorted(glob.glob(os.path.join(self.dir, "*.tfrecord.zz"))):
            raw = zlib.decompress(open(path, "rb").read())
            off = 0
            while off < len(raw):
                (n,) = struct.unpack_from("<Q", raw, off)
                (hc,) = struct.unpack_from("<I", raw, off + 8)
                assert hc == masked(crc32c_py(raw[off:off + 8]))
                data = raw[off + 12:off + 12 + n]
                (fc,) = struct.unpack_from("<I", raw, off + 12 + n)
                assert fc == masked(crc32c_py(data))
                ex = Example()
                ex.ParseFromString(data)
                out.append(ex.features.feature)
                off += 16 + n
        return out

    def close(self):
        self.L.p3host_tfrec_free(self.h)


def pass_game(n):
    return [PASS if i % 2 == 0 else -PASS for i in range(n)]


def one_hot(i):
    p = np.zeros(NUM_MOVES, np.float32)
    p[i] = 1
    return p


def fget(ex, key):
    return ex[key].float_list.value[0]


def bget(ex, key, dt):
    return np.frombuffer(ex[key].bytes_list.value[0], dt)


def exp_weighted(qs, m, lam):
    h = len(qs) - m - 1
    w = np.array([np.float32(lam) ** i for i in range(h + 1)], np.float64)
    sign = np.array([1 if i % 2 == 0 else -1 for i in range(h + 1)])
    return float((sign * w * np.asarray(qs[m:m + h + 1], np.float64)).sum() / w.sum())


def test_crc32c_known_answers(L):
    assert L.p3host_crc32c(b"123456789", 9) == 0xE3069283          # RFC 3720 check value
    assert L.p3host_crc32c(bytes(32), 32) == 0x8A9136AA             # RFC 3720 B.4: 32 zero bytes
    assert L.p3host_crc32c(bytes([0xFF] * 32), 32) == 0x62A8AB43    # 32 bytes of 0xFF
    blob = os.urandom(1000)
    assert L.p3host_crc32c(blob, len(blob)) == crc32c_py(blob)


def test_td_targets(L, tmp_path):
    """tf_recorder_test.cc "TD targets written correctly" (:150-190)."""
    qs, scores = [0.3, -0.1, 0.6], [2.0, -1.0, 4.0]
    r = Recorder(L, tmp_path)
    r.record(pass_game(3), root_q=qs, root_score=scores)
    assert r.flush() == 3
    exs = r.examples()
    assert len(exs) == 3
    for m, ex in enumerate(exs):
        for key, lam, src in (("q6", 5 / 6, qs), ("q16", 15 / 16, qs), ("q50", 49 / 50, qs),
                              ("q6_score", 5 / 6, scores), ("q16_score", 15 / 16, scores),
                              ("q50_score", 49 / 50, scores)):
            assert fget(ex, key) == pytest.approx(exp_weighted(src, m, lam), rel=1e-5, abs=1e-6)
    # horizon 0 at the last move collapses to the raw root value (:194-223)
    assert fget(exs[2], "q6") == pytest.approx(0.6, rel=1e-5)
    assert fget(exs[2], "q50_score") == pytest.approx(4.0, rel=1e-5)
    r.close()


def test_only_trainable_moves_and_no_file(L, tmp_path):
    """:414-488 — only trainable moves produce examples; none trainable -> no chunk."""
    pi = np.stack([one_hot(0), one_hot(1), one_hot(2)])
    for mask, want in (([0, 1, 0], [1]), ([1, 0, 0], [0]), ([0, 0, 1], [2])):
        d = tmp_path / ("m" + "".join(map(str, mask)))
        d.mkdir()
        r = Recorder(L, d)
        r.record(pass_game(3), pi=pi, trainable=mask)
        assert r.flush() == 1
        (ex,) = r.examples()
        assert list(np.flatnonzero(bget(ex, "pi", np.float32))) == want
        r.close()
    d = tmp_path / "none"
    d.mkdir()
    r = Recorder(L, d)
    r.record(pass_game(3), pi=pi, trainable=[0, 0, 0])
    assert r.flush() == 0 and os.listdir(d) == []
    r.close()


def test_two_games_one_flush_and_files(L, tmp_path):
    """:254-288 two games accumulated; chunk naming per cc/data/filename_format.h:11-37."""
    r = Recorder(L, tmp_path, gen=3, worker="w9")
    r.record(pass_game(2), pi=np.stack([one_hot(0)] * 2), trainable=[1, 0])
    r.record(pass_game(2), pi=np.stack([one_hot(1)] * 2), trainable=[1, 0])
    assert r.flush() == 2
    exs = r.examples()
    assert bget(exs[0], "pi", np.float32)[0] == 1 and bget(exs[1], "pi", np.float32)[1] == 1
    names = sorted(os.listdir(tmp_path))
    import re
    stem = re.fullmatch(r"(gen003_b000_g002_n00002_t\d+_w9)\.done", [n for n in names if n.endswith(".done")][0]).group(1)
    assert set(names) == {stem + e for e in (".done", ".stats", ".tfrecord.zz", ".visit_count")}
    assert open(tmp_path / (stem + ".visit_count")).read().splitlines()[2] == "Trainable Moves: 2"
    r.record(pass_game(2))
    r.flush()
    assert any("_b001_g001_n00002_" in n for n in os.listdir(tmp_path))   # batch number advanced
    r.close()


def test_fields(L, tmp_path):
    """score_margin sign (:290-313), pi_aux / pi_aux_dist (:315-347,:489-531), colour
    (:386-412), mcts_value_dist (:533-571), board planes and fixed fields."""
    B, W = 1, -1
    mv = [B * (0 * 19 + 3 + 1), W * (0 * 19 + 7 + 1)]       # B (0,3), W (0,7)
    dist = np.zeros((2, NUM_V_BUCKETS), np.uint32)
    dist[0, 10], dist[0, 40] = 3, 7
    r = Recorder(L, tmp_path)
    r.record(mv, pi=np.stack([one_hot(3), one_hot(7)]), dist=dist)
    assert r.flush() == 2
    e0, e1 = r.examples()
    assert fget(e0, "score_margin") == pytest.approx(-fget(e1, "score_margin"), rel=1e-5)
    assert fget(e0, "score_margin") == pytest.approx(-7.5)      # one stone each, komi 7.5
    assert bget(e0, "pi_aux", np.int16)[0] == 7 and bget(e1, "pi_aux", np.int16)[0] == 19 * 19   # kPassLoc sentinel
    d0, d1 = bget(e0, "pi_aux_dist", np.float32), bget(e1, "pi_aux_dist", np.float32)
    assert d0[7] == 1 and d0.sum() == 1 and not d1.any()
    assert bget(e0, "color", np.int8)[0] == B and bget(e1, "color", np.int8)[0] == W
    v = bget(e0, "mcts_value_dist", np.uint32)
    assert len(v) == NUM_V_BUCKETS and v[10] == 3 and v[40] == 7 and v.sum() == 10
    assert bget(e0, "bsize", np.uint8)[0] == 19 and fget(e0, "komi") == 7.5
    assert not bget(e0, "board", np.int8).any()                  # position BEFORE the move
    b1 = bget(e1, "board", np.int8)
    assert b1[3] == B and np.count_nonzero(b1) == 1
    assert list(bget(e0, "last_moves", np.int16)) == [-20] * 5   # five noop moves = {-1,-1}
    assert list(bget(e1, "last_moves", np.int16)) == [-20] * 4 + [3]
    assert len(bget(e0, "own", np.int8)) == 361
    for k in ("stones_atari", "stones_two_liberties", "stones_three_liberties", "stones_in_ladder"):
        assert len(bget(e1, k, np.int8)) == 361
    assert bget(e1, "stones_three_liberties", np.int8)[3] == B   # edge stone: three liberties
    r.close()


def test_policy_surprise_weighting(L, tmp_path):
    """freq_weight = 0.5 + 0.5 * kld / avg_kld: kld (3, 1) -> weights (1.25, 0.75): the first
    move is written once or twice, the second at most once (tf_recorder.cc:222-233)."""
    r = Recorder(L, tmp_path)
    for _ in range(40):
        r.record(pass_game(2), pi=np.stack([one_hot(0), one_hot(1)]), kld=[3.0, 1.0])
    n = r.flush()
    exs = r.examples()
    first = sum(1 for e in exs if bget(e, "pi", np.float32)[0] == 1)
    second = n - first
    assert 40 <= first <= 80 and 0 < second < 40
    assert abs(first - 50) <= 12 and abs(second - 30) <= 12     # E = 40*1.25, 40*0.75
    r.close()


def _stat(std, n_pre, sel=1.0, kld=0.1):
    return [0, 0.1, 0.1, 0.0, std, 1.0, 0.05, kld, kld, sel, 1.0, 128.0, n_pre]


def _stats_file(d):
    (p,) = glob.glob(os.path.join(str(d), "*.stats"))
    return open(p).read().splitlines()


def test_stats_file(L, tmp_path):
    """sel_mult_test.cc:104-149,151-189,287-317."""
    r = Recorder(L, tmp_path)
    r.record(pass_game(2), root_q=[0.1, 0.1], kld=[0.1, 0.1], stats=[_stat(0.15, 30, 1.5), _stat(0.20, 80, 0.5)])
    r.flush()
    lines = _stats_file(tmp_path)
    starts = [ln.split("=")[0].split(" ")[0] for ln in lines]
    for f in ("v_outcome_stddev", "v_outcome_stddev_adj", "freq_weight", "sel_mult_modifier", "expected_std.n30",
              "expected_std.n80", "sel_mult_mean"):
        assert f in starts
    assert starts.index("v_outcome_stddev_adj") < starts.index("expected_std.n30")
    kv = dict(ln.split("=") for ln in lines if "=" in ln)
    assert float(kv["expected_std.n30"]) == pytest.approx(0.15, rel=1e-3)
    assert float(kv["sel_mult_mean"]) == pytest.approx(1.0, rel=1e-3)
    assert lines[0] == "# percentiles: p01 p05 p10 ... p95 p99 (2 moves)"
    assert lines[1].split() == ["field", "p01"] + ["p%02d" % i for i in range(5, 100, 5)] + ["p99"]
    r.close()


def test_stats_percentiles(L, tmp_path):
    """sel_mult_test.cc:191-234 (kld 0.01..1.00 over 100 moves) and :236-271 (freq_weight)."""
    r = Recorder(L, tmp_path)
    for g in range(50):
        k0, k1 = (2 * g + 1) / 100.0, (2 * g + 2) / 100.0
        r.record(pass_game(2), root_q=[0.1, 0.1], kld=[k0, k1], stats=[_stat(0.15, 30, 1.0, k0), _stat(0.15, 30, 1.0, k1)])
    r.flush()
    row = [ln for ln in _stats_file(tmp_path) if ln.startswith("pre_kld ")][0].split()[1:]
    assert len(row) == 21
    p = list(map(float, row))
    # doctest::Approx(x).epsilon(e) accepts |a - x| < e * (1 + max(|a|, |x|))
    assert abs(p[0] - 0.01) < 0.02 * 1.02 and abs(p[10] - 0.50) < 0.05 * 1.5 and abs(p[20] - 0.99) < 0.02 * 1.99
    assert p[0] == pytest.approx(0.02) and p[20] == pytest.approx(0.99)   # index round(pct/100 * (n-1))
    r.close()
    d = tmp_path / "fw"
    d.mkdir()
    r = Recorder(L, d)
    r.record(pass_game(2), root_q=[0.1, 0.1], kld=[0.2, 0.2], stats=[_stat(0.15, 30, 1.0, 0.2)] * 2)
    r.flush()
    row = [ln for ln in _stats_file(d) if ln.startswith("freq_weight ")][0].split()[1:]
    assert all(float(x) == pytest.approx(1.0, rel=1e-3) for x in row)
    r.close()
