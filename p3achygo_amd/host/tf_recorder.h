// tf_recorder.h — training-chunk writer of the self-play host.
//
// Restates the reference's recorder::TfRecorder (cc/recorder/tf_recorder.cc:113-467) and
// what it depends on, without protobuf / abseil / TensorFlow:
//   * tf.Example wire format written by hand (cc/recorder/make_tf_example.h:20-79: every
//     feature is a one-element BytesList of the raw little-endian array, or a one-element
//     FloatList).  Map entries are emitted in key order (protobuf's deterministic mode);
//     the reference's own order is hash-map order, readers do not depend on it.
//   * TFRecord framing: uint64 length, masked CRC32C of the length, payload, masked CRC32C
//     of the payload (cc/data/tfrecord/record_writer.cc:214-228, crc32.h:38-43).
//   * one zlib stream per chunk, level 2, window 15, memLevel 9, Z_NO_FLUSH until Close
//     (tf_recorder.cc:268-279, compression_options.h:11-39).
//   * side files: .done, .visit_count, .stats (tf_recorder.cc:283-460), chunk names from
//     cc/data/filename_format.h:11-37.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "board.h"
#include "rng.h"
#include "search.h"

namespace p3 {

// ---- CRC32C (Castagnoli, reflected 0x82F63B78), masked as TFRecord wants it --------------
inline uint32_t Crc32c(const void* data, size_t n, uint32_t crc = 0) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  crc = ~crc;
  const uint8_t* p = (const uint8_t*)data;
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return ~crc;
}
inline uint32_t MaskedCrc32c(const void* data, size_t n) {   // crc32.h:38-43
  const uint32_t crc = Crc32c(data, n);
  return ((crc >> 15) | (crc << 17)) + 0xa282ead8u;
}

// ---- protobuf wire helpers ---------------------------------------------------------------
inline void PbVarint(std::string& o, uint64_t v) {
  while (v >= 0x80) {
    o.push_back((char)(v | 0x80));
    v >>= 7;
  }
  o.push_back((char)v);
}
inline void PbLenDelim(std::string& o, int field, const std::string& payload) {
  PbVarint(o, (uint64_t)(field << 3 | 2));
  PbVarint(o, payload.size());
  o += payload;
}
// Feature{ bytes_list = 1 { value = 1: bytes } }
inline std::string PbBytesFeature(const void* data, size_t n) {
  std::string list;
  PbLenDelim(list, 1, std::string((const char*)data, n));
  std::string feat;
  PbLenDelim(feat, 1, list);
  return feat;
}
// Feature{ float_list = 2 { value = 1: packed float } }
inline std::string PbFloatFeature(float v) {
  std::string packed((const char*)&v, 4), list, feat;
  PbLenDelim(list, 1, packed);
  PbLenDelim(feat, 2, list);
  return feat;
}
// Example{ features = 1 { feature = 1: map<string, Feature> } }
inline std::string PbExample(const std::map<std::string, std::string>& features) {
  std::string fs;
  for (const auto& kv : features) {
    std::string entry;
    PbLenDelim(entry, 1, kv.first);
    PbLenDelim(entry, 2, kv.second);
    PbLenDelim(fs, 1, entry);
  }
  std::string ex;
  PbLenDelim(ex, 1, fs);
  return ex;
}

// ---- per-move search records (cc/recorder/move_search_stats.h) ------------------------------
struct MoveSearchStats {
  bool sampled_raw_policy = false;
  float nn_q = 0, mcts_q = 0, nn_mcts_diff = 0, v_outcome_stddev = 0, prior_entropy = 0,
        nn_uncertainty = 0, kld = 0, pre_kld = 0, sel_mult_modifier = 0,
        sel_mult_modifier_weight = 0, visit_count = 0, visit_count_pre = 0;
};
struct MoveSearchRecord {
  float mcts_pi[kNumMoves] = {};
  uint8_t move_trainable = 0;
  float root_q_outcome = 0, root_score = 0, kld = 0;
  uint32_t mcts_value_dist[kNumVBuckets] = {};
  MoveSearchStats move_stats;
};

// make_tf_example.h:20-79.  `own` is the final ownership of the game, `next_move` the
// encoded (i*19+j, pass = 361) location of the following move.
inline std::string MakeTfExample(Board& board, const int16_t last_moves[5], const float* pi, int16_t pi_aux,
                                 const float* pi_aux_dist, const Game::Result& result,
                                 const uint32_t* mcts_value_dist, float q6, float q16, float q50,
                                 float q6_score, float q16_score, float q50_score, Color color, float komi) {
  std::map<std::string, std::string> f;
  const uint8_t bsize = kBoardLen;
  f["bsize"] = PbBytesFeature(&bsize, 1);
  f["board"] = PbBytesFeature(board.position().data(), kNumLocs);
  f["last_moves"] = PbBytesFeature(last_moves, 5 * sizeof(int16_t));
  Grid g = board.GetStonesInAtari();
  f["stones_atari"] = PbBytesFeature(g.data(), kNumLocs);
  g = board.GetStonesWithLiberties(2);
  f["stones_two_liberties"] = PbBytesFeature(g.data(), kNumLocs);
  g = board.GetStonesWithLiberties(3);
  f["stones_three_liberties"] = PbBytesFeature(g.data(), kNumLocs);
  g = board.GetLadderedStones();
  f["stones_in_ladder"] = PbBytesFeature(g.data(), kNumLocs);
  f["color"] = PbBytesFeature(&color, 1);
  f["komi"] = PbFloatFeature(komi);
  f["own"] = PbBytesFeature(result.ownership.data(), kNumLocs);
  f["pi"] = PbBytesFeature(pi, kNumMoves * sizeof(float));
  f["pi_aux"] = PbBytesFeature(&pi_aux, sizeof(int16_t));
  f["pi_aux_dist"] = PbBytesFeature(pi_aux_dist, kNumMoves * sizeof(float));
  f["mcts_value_dist"] = PbBytesFeature(mcts_value_dist, kNumVBuckets * sizeof(uint32_t));
  const float margin = color == kBlack ? result.bscore - result.wscore : result.wscore - result.bscore;
  f["score_margin"] = PbFloatFeature(margin);
  f["q6"] = PbFloatFeature(q6);
  f["q16"] = PbFloatFeature(q16);
  f["q50"] = PbFloatFeature(q50);
  f["q6_score"] = PbFloatFeature(q6_score);
  f["q16_score"] = PbFloatFeature(q16_score);
  f["q50_score"] = PbFloatFeature(q50_score);
  return PbExample(f);
}

// TFRecord stream compressed as ONE zlib stream (record_writer.cc:63-104,239-262).
class ZlibRecordWriter {
 public:
  explicit ZlibRecordWriter(const std::string& path, int level = 2) : f_(std::fopen(path.c_str(), "wb")) {
    std::memset(&z_, 0, sizeof z_);
    ok_ = f_ && deflateInit2(&z_, level, Z_DEFLATED, MAX_WBITS, 9, Z_DEFAULT_STRATEGY) == Z_OK;
  }
  ~ZlibRecordWriter() { Close(); }
  bool ok() const { return ok_; }
  void WriteRecord(const std::string& data) {
    char header[12], footer[4];
    const uint64_t n = data.size();
    std::memcpy(header, &n, 8);
    const uint32_t hc = MaskedCrc32c(header, 8), fc = MaskedCrc32c(data.data(), data.size());
    std::memcpy(header + 8, &hc, 4);
    std::memcpy(footer, &fc, 4);
    Deflate(header, 12, Z_NO_FLUSH);
    Deflate(data.data(), data.size(), Z_NO_FLUSH);
    Deflate(footer, 4, Z_NO_FLUSH);
  }
  void Close() {
    if (!f_) return;
    if (ok_) {
      Deflate(nullptr, 0, Z_FINISH);
      deflateEnd(&z_);
    }
    std::fclose(f_);
    f_ = nullptr;
  }

 private:
  void Deflate(const char* p, size_t n, int flush) {
    if (!ok_) return;
    z_.next_in = (Bytef*)p;
    z_.avail_in = (uInt)n;
    do {
      z_.next_out = out_;
      z_.avail_out = sizeof out_;
      const int rc = deflate(&z_, flush);
      if (rc != Z_OK && rc != Z_BUF_ERROR && rc != Z_STREAM_END) { ok_ = false; return; }
      std::fwrite(out_, 1, sizeof out_ - z_.avail_out, f_);
      if (rc == Z_STREAM_END) break;
    } while (z_.avail_out == 0 || z_.avail_in > 0);
  }
  FILE* f_;
  z_stream z_;
  bool ok_ = false;
  Bytef out_[1 << 16];
};

// p01, p05..p95, p99 of vals (tf_recorder.cc:24-40)
inline std::vector<float> ComputePercentiles(std::vector<float> vals) {
  std::sort(vals.begin(), vals.end());
  const int n = (int)vals.size();
  auto at = [&](float pct) {
    if (n == 0) return 0.0f;
    return vals[std::clamp((int)std::round(pct / 100.0f * (n - 1)), 0, n - 1)];
  };
  std::vector<float> out{at(1.0f)};
  for (int i = 5; i <= 95; i += 5) out.push_back(at((float)i));
  out.push_back(at(99.0f));
  return out;
}

class TfRecorder {
 public:
  TfRecorder(std::string dir, int gen, std::string worker_id, uint64_t seed = 0x7466726563ull)
      : dir_(std::move(dir)), gen_(gen), worker_(std::move(worker_id)), prob_(seed) {}

  // tf_recorder.cc:102-109
  void RecordGame(const Board& init_board, const Game& game, std::vector<MoveSearchRecord> infos) {
    if ((int)infos.size() != game.num_moves()) return;
    records_.push_back(Record{init_board, game, std::move(infos)});
  }
  int buffered() const { return (int)records_.size(); }
  int batch_num() const { return batch_; }
  const std::string& last_chunk() const { return last_chunk_; }

  struct Record {
    Board init_board;
    Game game;
    std::vector<MoveSearchRecord> infos;
  };
  // Detaches the buffered games (cheap; call under the caller's lock), to be written with
  // FlushRecords outside it: the replay of every game is the expensive part.
  std::vector<Record> TakeRecords() { return std::move(records_); }
  int Flush() { return FlushRecords(TakeRecords()); }

  // tf_recorder.cc:113-467.  Returns the number of examples written (0: no file).
  int FlushRecords(std::vector<Record> records) {
    size_t trainable_visits = 0, fast_visits = 0, n_trainable = 0, n_fast = 0;
    std::vector<std::string> examples;
    std::vector<MoveSearchStats> all_stats;
    std::vector<float> all_weights;
    for (Record& rec : records) {
      const Game& game = rec.game;
      const std::vector<MoveSearchRecord>& infos = rec.infos;
      size_t num_trainable = 0;
      float kld_sum = 0;
      for (const auto& mi : infos) {
        num_trainable += mi.move_trainable;
        if (mi.move_trainable) kld_sum += mi.kld;
        const size_t vc = (size_t)mi.move_stats.visit_count;
        (mi.move_trainable ? trainable_visits : fast_visits) += vc;
        (mi.move_trainable ? n_trainable : n_fast) += 1;
      }
      const float avg_kld = num_trainable > 0 ? kld_sum / num_trainable : 0.0f;
      auto freq_weight = [&](const MoveSearchRecord& mi) {
        return avg_kld == 0.0f ? 1.0f : 0.5f + 0.5f * (mi.kld / avg_kld);
      };
      Board board = rec.init_board;   // the game is replayed from its initial board
      const int nm = game.num_moves();
      for (int m = 0; m < nm; ++m) {
        int16_t last_moves[5];
        for (int off = 0; off < 5; ++off) last_moves[off] = EncodeLoc16(game.moves()[m + off].loc);
        const Move move = game.move(m);
        const MoveSearchRecord& mi = infos[m];
        if (mi.move_trainable) {
          const Loc next = m < nm - 1 ? game.move(m + 1).loc : kPassLoc;
          static const float kZeroPi[kNumMoves] = {};
          const float* aux_dist = m < nm - 1 ? infos[m + 1].mcts_pi : kZeroPi;
          auto short_term = [&](float lambda, int horizon, float* q, float* sc) {
            float N = 0, qs = 0, ss = 0;
            for (int i = 0; i <= horizon; ++i) N += std::pow(lambda, i);
            for (int i = 0; i <= horizon; ++i) {
              const float vm = (i % 2 == 0) ? 1.0f : -1.0f;
              qs += vm * std::pow(lambda, i) * infos[m + i].root_q_outcome;
              ss += vm * std::pow(lambda, i) * infos[m + i].root_score;
            }
            *q = qs / N;
            *sc = ss / N;
          };
          float q6, q16, q50, s6, s16, s50;
          short_term(5.0f / 6.0f, std::min(6, nm - m - 1), &q6, &s6);
          short_term(15.0f / 16.0f, std::min(16, nm - m - 1), &q16, &s16);
          short_term(49.0f / 50.0f, nm - m - 1, &q50, &s50);
          const std::string data =
              MakeTfExample(board, last_moves, mi.mcts_pi, EncodeLoc16(next), aux_dist, game.result(),
                            mi.mcts_value_dist, q6, q16, q50, s6, s16, s50, move.color, game.komi());
          const float fw = freq_weight(mi);   // policy surprise weighting
          for (int i = 0; i < (int)std::floor(fw); ++i) examples.push_back(data);
          if (prob_.Uniform() < fw - std::floor(fw)) examples.push_back(data);
        }
        if (move.loc == kPassLoc) board.Pass(move.color);
        else board.PlayMove(move.loc, move.color);
      }
      for (const auto& mi : infos) {
        const MoveSearchStats& s = mi.move_stats;
        if (!s.sampled_raw_policy && s.visit_count > 1.0f) {
          all_stats.push_back(s);
          all_weights.push_back(freq_weight(mi));
        }
      }
    }
    const int num_games = (int)records.size();
    records.clear();
    if (examples.empty()) return 0;

    const int num_records = (int)examples.size();
    const int ts = (int)std::chrono::duration_cast<std::chrono::seconds>(
                       std::chrono::steady_clock::now().time_since_epoch()).count();
    auto name = [&](const char* ext) {
      char buf[256];
      std::snprintf(buf, sizeof buf, "gen%03d_b%03d_g%03d_n%05d_t%d_%s.%s", gen_, batch_, num_games,
                    num_records, ts, worker_.c_str(), ext);
      return dir_ + "/" + buf;
    };
    last_chunk_ = name("tfrecord.zz");
    {
      ZlibRecordWriter w(last_chunk_, 2);
      for (const std::string& e : examples) w.WriteRecord(e);
      w.Close();
    }
    if (FILE* f = std::fopen(name("done").c_str(), "w")) std::fclose(f);
    if (FILE* f = std::fopen(name("visit_count").c_str(), "w")) {
      std::fprintf(f,
                   "Trainable Visits: %lu\nFast Visits: %lu\nTrainable Moves: %lu\nFast Moves: %lu\n"
                   "Visits Per Trainable Move: %lu\nVisits Per Fast Move: %lu\n",
                   trainable_visits, fast_visits, n_trainable, n_fast,
                   n_trainable > 0 ? trainable_visits / n_trainable : 0, n_fast > 0 ? fast_visits / n_fast : 0);
      std::fclose(f);
    }
    WriteStats(name("stats"), all_stats, all_weights);
    ++batch_;
    return num_records;
  }

  static int16_t EncodeLoc16(Loc l) { return (int16_t)(l.i * kBoardLen + l.j); }   // Loc -> int16, loc.h:24-26

 private:
  // percentile table + expected_std bins + sel_mult_mean, tf_recorder.cc:318-460
  static void WriteStats(const std::string& path, const std::vector<MoveSearchStats>& stats,
                         const std::vector<float>& weights) {
    FILE* f = std::fopen(path.c_str(), "w");
    if (!f) return;
    auto collect = [&](std::function<float(const MoveSearchStats&)> get) {
      std::vector<float> v;
      for (const auto& s : stats) {
        if (s.sampled_raw_policy) continue;
        const float x = get(s);
        if (x == 0.0f || !std::isfinite(x)) continue;
        v.push_back(x);
      }
      return v;
    };
    auto row = [&](const char* name, const std::vector<float>& p) {
      std::fprintf(f, "%-24s", name);
      for (float x : p) std::fprintf(f, " %9.6f", x);
      std::fprintf(f, "\n");
    };
    std::fprintf(f, "# percentiles: p01 p05 p10 ... p95 p99 (%d moves)\n", (int)stats.size());
    std::fprintf(f, "%-24s %9s", "field", "p01");
    for (int i = 5; i <= 95; i += 5) {
      char b[8];
      std::snprintf(b, sizeof b, "p%02d", i);
      std::fprintf(f, " %9s", b);
    }
    std::fprintf(f, " %9s\n", "p99");
    row("nn_q", ComputePercentiles(collect([](const MoveSearchStats& s) { return s.nn_q; })));
    row("mcts_q", ComputePercentiles(collect([](const MoveSearchStats& s) { return s.mcts_q; })));
    row("nn_mcts_diff", ComputePercentiles(collect([](const MoveSearchStats& s) { return s.nn_mcts_diff; })));
    row("v_outcome_stddev", ComputePercentiles(collect([](const MoveSearchStats& s) { return s.v_outcome_stddev; })));
    row("prior_entropy", ComputePercentiles(collect([](const MoveSearchStats& s) { return s.prior_entropy; })));
    row("nn_uncertainty", ComputePercentiles(collect([](const MoveSearchStats& s) { return s.nn_uncertainty; })));
    row("kld", ComputePercentiles(collect([](const MoveSearchStats& s) { return s.kld; })));
    row("pre_kld", ComputePercentiles(collect([](const MoveSearchStats& s) { return s.pre_kld; })));
    row("sel_mult_modifier", ComputePercentiles(collect([](const MoveSearchStats& s) { return s.sel_mult_modifier; })));
    row("visit_count", ComputePercentiles(collect([](const MoveSearchStats& s) { return s.visit_count; })));
    {
      std::vector<float> w;
      for (float x : weights)
        if (x != 0.0f && std::isfinite(x)) w.push_back(x);
      row("freq_weight", ComputePercentiles(w));
    }
    constexpr int kCap = 200;
    std::map<int, float> expected;
    {
      std::map<int, std::pair<float, int>> bins;
      float above = 0;
      int above_n = 0;
      for (const auto& s : stats) {
        if (s.sampled_raw_policy || s.v_outcome_stddev <= 0 || !std::isfinite(s.v_outcome_stddev) ||
            s.visit_count_pre <= 0)
          continue;
        const int n = (int)s.visit_count_pre;
        if (n >= kCap) {
          above += s.v_outcome_stddev;
          ++above_n;
        } else {
          bins[(n / 5) * 5].first += s.v_outcome_stddev;
          bins[(n / 5) * 5].second += 1;
        }
      }
      for (const auto& b : bins)
        if (b.second.second > 0) expected[b.first] = b.second.first / b.second.second;
      if (above_n > 0) expected[kCap] = above / above_n;
    }
    row("v_outcome_stddev_adj", ComputePercentiles(collect([&](const MoveSearchStats& s) {
          if (s.v_outcome_stddev <= 0 || s.visit_count_pre <= 0) return 0.0f;
          const int n = (int)s.visit_count_pre;
          auto it = expected.find(n >= kCap ? kCap : (n / 5) * 5);
          if (it == expected.end() || it->second <= 0.0f) return 0.0f;
          return s.v_outcome_stddev / it->second;
        })));
    for (const auto& e : expected) std::fprintf(f, "expected_std.n%d=%f\n", e.first, e.second);
    float sm_sum = 0, sm_cnt = 0;
    for (const auto& s : stats) {
      if (s.sampled_raw_policy || !std::isfinite(s.sel_mult_modifier)) continue;
      sm_sum += s.sel_mult_modifier_weight * s.sel_mult_modifier;
      sm_cnt += s.sel_mult_modifier_weight;
    }
    std::fprintf(f, "sel_mult_mean=%f\n", sm_cnt > 0 ? sm_sum / sm_cnt : 1.0f);
    std::fclose(f);
  }

  std::string dir_;
  int gen_;
  std::string worker_;
  Probability prob_;
  std::vector<Record> records_;
  int batch_ = 0;
  std::string last_chunk_;
};

}  // namespace p3
