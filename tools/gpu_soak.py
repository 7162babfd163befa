"""Soak run: self-play on the HIP engine with recording for a while; reports throughput per
interval and checks that outputs stay finite (a stall, a fault or NaNs would show here)."""
import os, sys, tempfile, time, glob, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from p3achygo_amd import host_api, netspec
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = netspec.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "b12c256btl3"]
d = tempfile.mkdtemp()
path = os.path.join(d, "n.p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
rec = os.path.join(d, "rec")
os.makedirs(rec)
host_api.set_recorder(rec, gen=1, worker_id="soak", flush_interval=64)
host_api.set_groups(3)
t0 = time.time()
total = 0
max_moves = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for i in range(1):
    st = host_api.selfplay_run(path, 3072, 16, secs, default_n=32, default_k=5, selected_n=64, selected_k=8,
                               max_moves=max_moves, warmup_batches=2, seed=100 + i)
    total += st.positions
    print(f"[{time.time()-t0:6.1f}s] {st.positions/st.seconds:,.0f} pos/s  games={st.games} moves={st.moves} "
          f"cache_hits={st.cache_hits} reuse_added={host_api.last_run_counters()[0]} examples={host_api.last_run_counters()[1]}",
          flush=True)
chunks = glob.glob(os.path.join(rec, "chunks", "*.tfrecord.zz"))
nbytes = sum(len(zlib.decompress(open(c, "rb").read())) for c in chunks)
print("chunks", len(chunks), "decompressed MB", nbytes / 1e6, "sgf files", len(glob.glob(os.path.join(rec, "sgf", "*.sgf"))))
