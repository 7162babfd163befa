"""BASELINE configs[2] as stated (1024 games, batch 1024, ONE game group) with one lane and with two lanes at several
depths (host_api.set_lanes), on the GPU: positions/s, batch fill, host share.  Usage: python tools/gpu_lanes_c3.py [steps]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p3achygo_amd import host_api, netspec

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = netspec.CONFIGS["b12c256btl3"]
path = os.path.join(tempfile.mkdtemp(prefix="p3lanes"), "net.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
threads = max(2, min(16, len(os.sched_getaffinity(0))))
host_api.set_advance_limit(400)
host_api.set_step_limit(steps)
CASES = ((1, 1, 1, 1024), (1, 2, 2, 1024), (1, 2, 3, 1024), (1, 2, 4, 1024), (2, 1, 1, 2048), (2, 2, 4, 2048), (8, 1, 1, 8192))
if len(sys.argv) > 2:
    CASES = tuple(tuple(int(v) for v in a.split(",")) for a in sys.argv[2:])
for groups, lanes, depth, games in CASES:
    host_api.set_groups(groups)
    host_api.set_lanes(lanes, depth)
    st = host_api.selfplay_run(path, games, threads, 0.0, default_n=32, default_k=5, selected_n=32, selected_k=5,
                               warmup_batches=8, seed=177)
    d = host_api.last_first_game_digests()
    print(f"groups {groups} lanes {lanes} depth {depth} games {games}: {st.positions / st.seconds:9.0f} positions/s  "
          f"fill {st.positions / max(st.batches, 1) / 1024:.3f}  batches {st.batches}  ms/batch {1e3 * st.seconds / st.batches:.3f}  "
          f"host share {st.host_seconds / st.seconds:.3f}  evals/move {st.positions / max(st.moves, 1):.2f}  "
          f"handed over (phases, games) {host_api.last_handed_over()}", flush=True)
