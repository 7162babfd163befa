#!/usr/bin/env python3
"""A/B of the hand-scheduled block kernel k_blockw (P3HIP_BLOCKW=1) against the shipped HIP kernels, on the GPU box:
outputs of both builds on the same positions (max |difference| of the raw head outputs), then forward-pass timing.
Each engine runs in a child process (the switch is read at engine creation), under a timeout.

    python tools/gpu_blockw_ab.py [parity|time|stamps] ...
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, os, tempfile, time
sys.path.insert(0, %r)
import numpy as np
from p3achygo_amd import engine, features, netspec
mode, name, batch, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
cfg = netspec.CONFIGS[name]
path = os.path.join(tempfile.mkdtemp(), "n.p3w")
Wts = netspec.generate_weights(cfg, randomize=True)
if os.environ.get("P3_AB_ZERO"):      # every MFMA operand zero: the chip holds its full clock (DESIGN.md section 4, the clock)
    Wts = {k: (np.zeros_like(v) if not k.endswith(".var") else v) for k, v in Wts.items()}
netspec.save_p3w(path, cfg, Wts)
pos = features.random_positions(batch, seed=5, n_games=9)
eng = engine.HipEngine(path, batch)
eng.load_all(pos)
if mode == "parity":
    eng.RunInference()
    np.save(out, np.stack([eng.get_raw(i) for i in range(batch)]))
elif mode == "xdiff":
    eng.upload()
    eng.forward_resident(batch)
    eng.sync()
    np.save(out, eng.debug_x(batch, cfg.channels))
elif mode == "time":
    eng.upload()
    for _ in range(5):
        eng.forward_resident(batch)
    eng.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(30):
            eng.forward_resident(batch)
        eng.sync()
        best = min(best, (time.perf_counter() - t0) / 30)
    ms, fl, kn = eng.time_trunk_kernel(batch, 10)
    print("RESULT", name, batch, "forward_ms %%.4f" %% (best * 1e3), "trunk_launch_ms %%.4f" %% ms, "flops %%.4g" %% fl, kn,
          "frac %%.4f" %% (fl / (ms * 1e-3) / 2.5e15), flush=True)
elif mode == "stamps":
    eng.upload()
    for _ in range(3):
        eng.forward_resident(batch)
    eng.sync()
    np.save(out, eng.blockw_stamps())
eng.close()
"""


def run(mode, name, batch, env_extra, out, timeout=180):
    env = dict(os.environ)
    for k in ("P3HIP_BLOCKW", "P3HIP_BLOCKW_DIAG", "P3HIP_NO_BFUSE", "P3_AB_ZERO"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT, mode, name, str(batch), out], env=env, capture_output=True,
                       text=True, timeout=timeout)
    if r.returncode != 0:
        print("FAILED", mode, name, batch, env_extra, r.stderr[-1500:], flush=True)
        return None
    return r.stdout


def main():
    import numpy as np
    what = sys.argv[1] if len(sys.argv) > 1 else "parity"
    tmp = tempfile.mkdtemp()
    if what == "parity":
        cases = [("test_b3c256btl1", 5), ("test_b5c256btl2_i2", 37), ("test_b10c256btl1_i2", 11), ("b12c256btl3", 7), ("b12c256btl3", 300)]
        ok = True
        for name, batch in cases:
            a, b = os.path.join(tmp, "a.npy"), os.path.join(tmp, "b.npy")
            if run("parity", name, batch, {"P3HIP_NO_BFUSE": "1"}, a) is None or run("parity", name, batch, {"P3HIP_BLOCKW": "1"}, b) is None:
                ok = False
                break
            x, y = np.load(a), np.load(b)
            d = np.abs(x - y)
            print(f"PARITY {name} batch {batch}: max |blockw - hip| = {d.max():.3e} (logits {d[:, :1887].max():.3e}), nan {int(np.isnan(y).sum())}, "
                  f"worst position {int(d.max(axis=1).argmax())}, |ref| max {np.abs(x).max():.2f}", flush=True)
            if not (d.max() < 2e-2):
                ok = False
                print("   per-position max:", np.round(d.max(axis=1)[:16], 4), flush=True)
        sys.exit(0 if ok else 1)
    if what == "xdiff":
        # x after the first run of blocks (P3HIP_DEBUG_STOP_BLOCK = first broadcast block), both builds
        name = sys.argv[2] if len(sys.argv) > 2 else "test_b10c256btl1_i2"
        batch = int(sys.argv[3]) if len(sys.argv) > 3 else 3
        stop = sys.argv[4] if len(sys.argv) > 4 else "1"
        a, b = os.path.join(tmp, "a.npy"), os.path.join(tmp, "b.npy")
        if run("xdiff", name, batch, {"P3HIP_NO_BFUSE": "1", "P3HIP_DEBUG_STOP_BLOCK": stop}, a) is None: sys.exit(1)
        if run("xdiff", name, batch, {"P3HIP_BLOCKW": "1", "P3HIP_DEBUG_STOP_BLOCK": stop}, b) is None: sys.exit(1)
        c = os.path.join(tmp, "c.npy")
        if run("xdiff", name, batch, {"P3HIP_NO_BFUSE": "1", "P3HIP_DEBUG_STOP_BLOCK": "0"}, c) is None: sys.exit(1)
        x, y, xin = np.load(a), np.load(b), np.load(c)          # [n][C][361]
        ok = ~np.isnan(y)
        print("|new - x_in| max over non-NaN", np.abs(y - xin)[ok].max(), " |ref - x_in| max", np.abs(x - xin).max())
        d = np.abs(x - y)
        d = np.where(np.isnan(d), 1e9, d)
        print("x after block(s): max diff", d.max(), "nan count", int(np.isnan(y).sum()), "ref max", np.abs(x).max())
        bad = d > 2e-2
        print("bad fraction", bad.mean())
        print("bad by position", bad.reshape(batch, -1).mean(axis=1))
        print("bad by channel block of 16:", np.round(bad.mean(axis=(0, 2)).reshape(-1, 16).mean(axis=1), 3).tolist())
        print("bad by channel within 16 (mod 16):", np.round(bad.mean(axis=(0, 2)).reshape(-1, 16).mean(axis=0), 3).tolist())
        rows = bad.mean(axis=(0, 1)).reshape(19, 19)
        print("bad by board row y:", np.round(rows.mean(axis=1), 3).tolist())
        print("bad by board column x:", np.round(rows.mean(axis=0), 3).tolist())
        np.set_printoptions(linewidth=200, precision=3, suppress=True)
        print("sample ref  [pos 0, ch 0..7, loc 0..7]\n", x[0, :8, :8])
        print("sample new  [pos 0, ch 0..7, loc 0..7]\n", y[0, :8, :8])
        print("sample ref  [pos 0, ch 128..131, loc 180..187]\n", x[0, 128:132, 180:188])
        print("sample new  [pos 0, ch 128..131, loc 180..187]\n", y[0, 128:132, 180:188])
        sys.exit(0)
    if what in ("time", "timezero"):
        name = sys.argv[2] if len(sys.argv) > 2 else "b12c256btl3"
        batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
        zero = {"P3_AB_ZERO": "1"} if what == "timezero" else {}
        for rnd in range(2):
            for label, env in (("joined", {}), ("blocks_only", {"P3HIP_NO_BFUSE": "1"}), ("blockw", {"P3HIP_BLOCKW": "1"})):
                out = run("time", name, batch, dict(env, **zero), "-")
                print(("zero-data " if zero else "") + label, (out or "").strip().splitlines()[-1] if out else "FAILED", flush=True)
    if what == "stamps":
        name = sys.argv[2] if len(sys.argv) > 2 else "b12c256btl3"
        batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
        out = os.path.join(tmp, "s.npy")
        if run("stamps", name, batch, {"P3HIP_BLOCKW": "1", "P3HIP_BLOCKW_DIAG": "1"}, out) is None:
            sys.exit(1)
        st = np.load(out).astype(np.int64)      # [wg 8][block 16][wave 4][24]

        blk = st[:, 1]                           # the run's second block: steady state
        n = int((blk[0, 0, :21] != 0).sum())
        d = np.diff(blk[:, :, :n], axis=2)       # [wg][wave][section]
        print("sections (cycles, median over 8 workgroups x 4 waves), block 1 of run 0:")
        print(np.median(d.reshape(-1, n - 1), axis=0).astype(int).tolist())
        print("per wave of workgroup 0:")
        for w in range(4):
            print(w, d[0, w].tolist())
        print("block total:", int(np.median(blk[:, :, n - 1] - blk[:, :, 0])))
        rt = (blk[:, :, 22] - blk[:, :, 21]).astype(np.float64) * 10e-9
        cyc = (blk[:, :, 23] - blk[:, :, 0]).astype(np.float64)
        print("in-kernel clock over the block: %.3f GHz; block wall %.1f us" % (float(np.median(cyc / rt)) / 1e9, float(np.median(rt)) * 1e6))


if __name__ == "__main__":
    main()
