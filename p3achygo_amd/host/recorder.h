// recorder.h — SGF output of finished games, byte-compatible with the reference.
//   SgfSerializer::Serialize     cc/sgf/sgf_serializer.cc:35-96  ("(;FF[4]GM[1]" + props + moves + ")")
//   header properties            cc/recorder/sgf_recorder.cc:98-104 (KM, RE, PB, PW in this order)
//   move coordinates             sgf_serializer.cc:27-32: ROW letter then COLUMN letter, pass = ""
//   result string                sgf_serializer.cc:12-25 ("B+%g" / "W+%g" / "B+R" / "W+R" / "?")
//   batch files                  sgf_recorder.cc:266-326 + cc/data/filename_format.h:27-31:
//                                gen%03d_b%03d_g%03d_%s.sgf, one game per line, then a .done file
// TFRecord chunks (cc/recorder/tf_recorder.cc) are not written yet.
#pragma once
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "board.h"

namespace p3 {

inline std::string SgfResultString(const Game::Result& r) {
  char buf[64];
  if (r.winner == kBlack) {
    if (r.by_resign) return "B+R";
    snprintf(buf, sizeof buf, "B+%g", r.bscore - r.wscore);
    return buf;
  }
  if (r.winner == kWhite) {
    if (r.by_resign) return "W+R";
    snprintf(buf, sizeof buf, "W+%g", r.wscore - r.bscore);
    return buf;
  }
  return "?";
}

inline std::string SgfGameString(float komi, const Game::Result& result, const std::vector<Move>& moves,
                                 size_t first_move, const std::string& b_name, const std::string& w_name) {
  static const char kCoords[] = "abcdefghijklmnopqrst";
  char buf[64];
  std::string s = "(;FF[4]GM[1]";
  snprintf(buf, sizeof buf, "KM[%g]", komi);
  s += buf;
  s += "RE[" + SgfResultString(result) + "]PB[" + b_name + "]PW[" + w_name + "]";
  for (size_t i = first_move; i < moves.size(); ++i) {
    const Move& m = moves[i];
    if (m.color != kBlack && m.color != kWhite) continue;
    s += m.color == kBlack ? ";B[" : ";W[";
    if (m.loc != kPassLoc) {
      s += kCoords[m.loc.i];
      s += kCoords[m.loc.j];
    }
    s += "]";
  }
  s += ")";
  return s;
}

inline std::string SgfGameString(const Game& g, const std::string& b_name, const std::string& w_name) {
  return SgfGameString(g.komi(), g.result(), g.moves(), Game::kMoveOffset, b_name, w_name);
}

// Buffers serialized games and writes one batch file per Flush (SgfRecorderImpl::Flush).
class SgfRecorder {
 public:
  SgfRecorder(std::string dir, int gen, std::string worker_id) : dir_(std::move(dir)), gen_(gen), worker_(std::move(worker_id)) {}
  void RecordGame(const std::string& sgf) {
    std::lock_guard<std::mutex> l(mu_);
    buf_ += sgf;
    buf_ += "\n";
    ++games_;
  }
  int buffered() const { return games_; }
  // returns the path written ("" if nothing was buffered)
  std::string Flush() {
    std::lock_guard<std::mutex> l(mu_);
    if (games_ == 0) return "";
    char name[256], done[256];
    snprintf(name, sizeof name, "gen%03d_b%03d_g%03d_%s.sgf", gen_, batch_, games_, worker_.c_str());
    snprintf(done, sizeof done, "gen%03d_b%03d_g%03d_%s.done", gen_, batch_, games_, worker_.c_str());
    const std::string path = dir_ + "/" + name;
    if (FILE* f = fopen(path.c_str(), "w")) {
      fwrite(buf_.data(), 1, buf_.size(), f);
      fclose(f);
    }
    if (FILE* f = fopen((dir_ + "/" + done).c_str(), "w")) fclose(f);
    buf_.clear();
    games_ = 0;
    ++batch_;
    return path;
  }

 private:
  std::string dir_;
  int gen_;
  std::string worker_;
  std::mutex mu_;
  std::string buf_;
  int games_ = 0, batch_ = 0;
};

}  // namespace p3
