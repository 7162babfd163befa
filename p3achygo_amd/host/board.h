// board.h — Go rules engine of the self-play host.
//
// Behaviour restates the reference's cc/game/board.{h,cc} (cited per method) and must be
// bit-exact in everything observable: stones, move legality (suicide, positional superko,
// pass-alive prohibition with the reference's *update points*), captures, area scores and
// ownership with Benson pass-alive dead-stone removal, the 1/2/3-liberty planes and the
// ladder plane.  The data structures are new and built for cheap copies (the search makes
// one Board copy per playout): stones of a group form a circular linked list, per-group
// exact liberty counts are maintained incrementally, and the superko history is a small
// open-addressed table of 64-bit Zobrist keys inside the object (no heap).
//
// Hash values are deterministic (fixed-seed table) and NOT comparable with the reference's,
// whose Zobrist table is time-seeded (cc/game/zobrist.cc:10).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace p3 {

constexpr int kBoardLen = 19;
constexpr int kNumLocs = kBoardLen * kBoardLen;
constexpr int kNumMoves = kNumLocs + 1;          // constants::kMaxMovesPerPosition
constexpr int kPassEncoding = kNumLocs;          // constants::kPassMoveEncoding
constexpr int kNumPassesBeforeBensons = 3;       // constants.h:75
using Color = int8_t;
constexpr Color kEmpty = 0, kBlack = 1, kWhite = -1;
inline Color Opp(Color c) { return (Color)-c; }

struct Loc {
  int i, j;
  bool operator==(const Loc& o) const { return i == o.i && j == o.j; }
  bool operator!=(const Loc& o) const { return !(*this == o); }
};
constexpr Loc kNoopLoc{-1, -1};                  // cc/game/loc.h:46
constexpr Loc kPassLoc{19, 0};                   // cc/game/loc.h:47
inline int Idx(Loc l) { return l.i * kBoardLen + l.j; }
inline Loc AsLoc(int idx) { return Loc{idx / kBoardLen, idx % kBoardLen}; }

enum class MoveStatus : uint8_t {                // cc/game/board.h:50-58
  kValid, kUnknownColor, kOutOfBounds, kLocNotEmpty, kPassAliveRegion, kSelfCapture,
  kRepeatedPosition,
};
inline bool MoveOk(MoveStatus s) { return s == MoveStatus::kValid; }

using Grid = std::array<Color, kNumLocs>;

struct Scores {                                  // cc/game/board.h:71-75
  float black_score, white_score;
  Grid ownership;
};

// Positional-superko history.  A Board keeps the keys of the positions it added itself in
// a short inline list and shares everything older through an immutable open-addressed
// table, so copying a Board (one per playout, one per ladder-reader node) costs ~3 KB.
struct SeenTable {
  static constexpr int kCap = 4096;   // >= 2 * (kMaxGameLen + deepest read), power of 2
  uint64_t slot[kCap] = {};
  int count = 0;
  bool Contains(uint64_t h) const;
  void Insert(uint64_t h);
};

class Board {
 public:
  explicit Board(float komi = 7.5f, bool prohibit_pass_alive = true);
  Board(const Board& o) { *this = o; }
  Board& operator=(const Board& o) {
    if (this == &o) return *this;
    std::memcpy(static_cast<void*>(&stones_), static_cast<const void*>(&o.stones_),
                reinterpret_cast<const char*>(&o.local_) - reinterpret_cast<const char*>(&o.stones_));
    std::memcpy(local_, o.local_, sizeof(uint64_t) * o.local_n_);
    base_ = o.base_;
    return *this;
  }
  // handicap constructor, cc/game/board.cc:444-470 (2..4 stones)
  Board(int handicap, float komi);

  int at(int i, int j) const { return stones_[i * kBoardLen + j]; }
  Color at(int idx) const { return stones_[idx]; }
  float komi() const { return komi_; }
  void SetKomi(float k) { komi_ = k; }
  uint64_t hash() const { return hash_; }
  int move_count() const { return move_count_; }
  const Grid& position() const { return stones_; }
  const Grid& pass_alive() const { return pass_alive_; }
  int consecutive_passes() const { return consecutive_passes_; }

  bool IsValidMove(Loc loc, Color color) const;                  // board.cc:492-498
  bool IsGameOver() const { return consecutive_passes_ == 2; }  // board.cc:500
  bool IsAllPassAlive();                                         // board.cc:502-507
  MoveStatus PlayMove(Loc loc, Color color);                     // board.cc:512-560
  MoveStatus Pass(Color color);                                  // board.cc:562-572
  // legality only; fills *new_hash when valid                   // board.cc:574-625
  MoveStatus PlayMoveDry(Loc loc, Color color, uint64_t* new_hash = nullptr) const;
  Scores GetScores();                                            // board.cc:627-650
  void CalculatePassAliveRegions();                              // board.cc:223-233
  void CalculatePassAliveRegionForColor(Color color);            // board.cc:246-275

  Grid GetStonesWithLiberties(int liberties) const;              // board.cc:670-690
  Grid GetStonesInAtari() const { return GetStonesWithLiberties(1); }
  Grid GetLadderedStones() const;                                // board.cc:692-899

  // liberties of the group at a stone (0 for an empty point)
  int LibertiesAt(int idx) const { return gid_[idx] < 0 ? 0 : libs_[gid_[idx]]; }
  int GroupIdAt(int idx) const { return gid_[idx]; }

  // Test hook mirroring GroupTracker::NewGroup/AddToGroup use in the reference's
  // PassAliveTest (board_test.cc:507-860): put a stone with no capture logic.
  void PlaceRaw(Loc loc, Color color);

  bool SamePosition(const Board& o) const { return stones_ == o.stones_; }
  bool IsEmpty() const {   // board.h:305-308
    for (Color c : stones_) if (c != kEmpty) return false;
    return true;
  }

 private:
  friend struct LadderSolver;
  int EmptyNeighbors(int idx) const;
  void RemoveGroup(int head);
  void AddStone(int idx, Color color);
  float ScoreAndOwnership(Color color, Grid& ownership) const;   // board.cc:917-988
  bool SeenContains(uint64_t h) const;
  void SeenInsert(uint64_t h);

  Grid stones_{};
  std::array<int16_t, kNumLocs> gid_;     // group head index, -1 when empty
  std::array<int16_t, kNumLocs> next_;    // circular list of a group's stones
  std::array<int16_t, kNumLocs> libs_;    // exact liberty count, valid at head
  Grid pass_alive_{};
  int move_count_ = 0, consecutive_passes_ = 0, passes_ = 0;
  int b_prisoners_ = 0, w_prisoners_ = 0;
  float komi_;
  bool prohibit_pass_alive_;
  uint64_t hash_;
  int local_n_ = 0;
  static constexpr int kLocalCap = 96;
  uint64_t local_[kLocalCap];             // keys added by this object since the last flush
  std::shared_ptr<const SeenTable> base_;  // older keys, shared between copies (may be null)
};

// ladder read-out statistics since process start: calls, nodes, max nodes of one call, budget hits
void LadderStats(long out[4]);
// Work bound of one ladder read-out in nodes; 0 (the default) = none, i.e. the reference's
// behaviour, bounded by depth 300 only (cc/game/board.cc:780-783).  A positive budget is the
// self-play throughput mode: a read-out that exhausts it reads "not laddered" (counted in
// LadderStats), which can differ from the reference on chaotic positions.  Process-wide.
void SetLadderNodeBudget(long nodes);
long LadderNodeBudget();

// cc/game/game.{h,cc}: a board plus the move list (five leading noop moves) and result.
struct Move {
  Color color;
  Loc loc;
};

class Game {
 public:
  static constexpr int kMoveOffset = 5;
  struct Result {
    Color winner = kEmpty;
    float bscore = 0, wscore = 0;
    bool by_resign = false;
    Grid ownership{};
  };
  explicit Game(float komi = 7.5f, bool prohibit_pass_alive = true);
  // restart from a stored position: game.cc:9-17 (moves_ = the five last moves)
  Game(const Board& board, const Move last_moves[5], int init_mv_num);
  int init_mv_num() const { return init_mv_num_; }
  const Board& board() const { return board_; }
  Board& mutable_board() { return board_; }
  int num_moves() const { return (int)moves_.size() - kMoveOffset; }
  Move move(int n) const { return moves_[n + kMoveOffset]; }
  const std::vector<Move>& moves() const { return moves_; }
  float komi() const { return board_.komi(); }
  bool IsGameOver() const { return board_.IsGameOver(); }
  bool IsValidMove(Loc loc, Color color) const { return board_.IsValidMove(loc, color); }
  bool PlayMove(Loc loc, Color color);            // game.cc:49-56 (passes are recorded too)
  Scores GetScores() { return board_.GetScores(); }
  void WriteResult();                             // game.cc:69-79
  const Result& result() const { return result_; }

 private:
  Board board_;
  std::vector<Move> moves_;
  Result result_;
  int init_mv_num_ = 0;
};


// What the search needs of a game: the board, the last five moves (network input) and the
// move count — cheap to copy once per playout (the reference copies the whole Game with its
// move vector and hash set: gumbel.cc:420).
struct Position {
  Board board;
  Move last[5];
  int num_moves = 0;
  Position() { for (auto& m : last) m = Move{kEmpty, kNoopLoc}; }
  explicit Position(const Game& g) : board(g.board()), num_moves(g.num_moves()) {
    for (int i = 0; i < 5; ++i) last[i] = g.moves()[g.moves().size() - 5 + i];
  }
  float komi() const { return board.komi(); }
  bool IsGameOver() const { return board.IsGameOver(); }
  bool IsValidMove(Loc loc, Color c) const { return board.IsValidMove(loc, c); }
  bool PlayMove(Loc loc, Color c) {
    if (!MoveOk(board.PlayMove(loc, c))) return false;
    for (int i = 0; i < 4; ++i) last[i] = last[i + 1];
    last[4] = Move{c, loc};
    ++num_moves;
    return true;
  }
  Scores GetScores() { return board.GetScores(); }
};

}  // namespace p3
