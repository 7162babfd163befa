"""Thread-per-game evaluation match (the reference's own shape: two NNInterfaces, threaded search) on the C = 384 trunks,
with the NN cache on the host (per-thread LRUs) and in the engines' HBM tables."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from p3achygo_amd import host_api, netspec
d = tempfile.mkdtemp()
paths = []
for name in ("b10c384nbt", "b14c384btl3"):
    cfg = netspec.CONFIGS[name]
    p = os.path.join(d, name + ".p3w")
    netspec.save_p3w(p, cfg, netspec.generate_weights(cfg))
    paths.append(p)
games = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for label, log2 in (("host LRU caches", 0), ("HBM tables (2^16 entries per engine)", 16)):
    host_api.set_device_nn_cache(log2)
    t0 = time.time()
    st = host_api.eval_match_threads(paths[0], paths[1], num_games=games, visits_per_move=64, threads_per_game=4,
                                     max_moves=40, cache_size=(1 << 16) if log2 == 0 else 0, seed=3)
    dt = time.time() - t0
    print(f"{label}: games={st.games} moves={st.moves} visits={st.visits} seconds={dt:.1f} visits/s={st.visits/dt:.0f} "
          f"cur/cand/draw={st.cur_wins}/{st.cand_wins}/{st.draws} device lookups={host_api.device_nn_cache_lookups()} hits={host_api.device_nn_cache_hits()}", flush=True)
host_api.set_device_nn_cache(0)
