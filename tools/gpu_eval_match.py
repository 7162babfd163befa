"""Evaluation-match throughput (config 5): b10c384nbt (cur) vs b14c384btl3 (cand), parallel
search with 8 leaves per round, 128 visits per move."""
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
games = int(sys.argv[1]) if len(sys.argv) > 1 else 128
moves = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dc = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # log2 entries of the engines' HBM NN cache, 0 = host caches only
host_api.set_device_nn_cache(dc)
st = host_api.eval_match(paths[0], paths[1], num_games=games, visits_per_move=128, leaves_per_round=8,
                         max_moves=moves, num_threads=16, seed=1)
if dc:
    print(f"HBM cache 2^{dc}: lookups={host_api.device_nn_cache_lookups()} hits={host_api.device_nn_cache_hits()}")
print(f"games={st.games} moves={st.moves} visits={st.visits} positions={st.positions} batches={st.batches} "
      f"collisions={st.collisions} seconds={st.seconds:.2f} positions/s={st.positions/st.seconds:.0f} "
      f"avg batch={st.positions/max(st.batches,1):.0f} cur/cand/draw={st.cur_wins}/{st.cand_wins}/{st.draws}")
