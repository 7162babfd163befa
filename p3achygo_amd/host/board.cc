// board.cc — see board.h.  Each routine names the reference lines whose behaviour it keeps.
#include "board.h"

#ifdef __AVX2__
#include <immintrin.h>
#endif

#include <atomic>

#include <algorithm>

#include "rng.h"

namespace p3 {
namespace {

struct Nbrs {
  int8_t n[kNumLocs];
  int16_t p[kNumLocs][4];
  Nbrs() {
    for (int idx = 0; idx < kNumLocs; ++idx) {
      int i = idx / kBoardLen, j = idx % kBoardLen, k = 0;
      if (i > 0) p[idx][k++] = (int16_t)(idx - kBoardLen);
      if (j > 0) p[idx][k++] = (int16_t)(idx - 1);
      if (i < kBoardLen - 1) p[idx][k++] = (int16_t)(idx + kBoardLen);
      if (j < kBoardLen - 1) p[idx][k++] = (int16_t)(idx + 1);
      n[idx] = (int8_t)k;
    }
  }
};
const Nbrs kNbr;

// Fixed-seed Zobrist keys, one per (point, state) with state = colour + 1
// (cc/game/board.cc:48 ZobristState).
struct ZobristTable {
  uint64_t k[kNumLocs][3];
  uint64_t empty_board;
  ZobristTable() {
    PRng rng(0x5a6f627269737431ull, 0x70336163ull);
    for (int p = 0; p < kNumLocs; ++p)
      for (int s = 0; s < 3; ++s) k[p][s] = rng.next64();
    empty_board = 0;
    for (int p = 0; p < kNumLocs; ++p) empty_board ^= k[p][kEmpty + 1];
  }
};
const ZobristTable kZob;

// epoch-stamped scratch marks (per thread, never part of a Board)
struct Marks {
  uint32_t stamp[kNumLocs] = {};
  uint32_t epoch = 0;
  void next() {
    if (++epoch == 0) {
      std::memset(stamp, 0, sizeof stamp);
      epoch = 1;
    }
  }
  bool test_and_set(int idx) {
    if (stamp[idx] == epoch) return true;
    stamp[idx] = epoch;
    return false;
  }
  bool test(int idx) const { return stamp[idx] == epoch; }
};
thread_local Marks g_marks, g_marks2;

inline uint64_t mix(uint64_t h) {
  h ^= h >> 29;
  h *= 0xbf58476d1ce4e5b9ull;
  h ^= h >> 32;
  return h;
}

}  // namespace

Board::Board(float komi, bool prohibit_pass_alive)
    : komi_(komi), prohibit_pass_alive_(prohibit_pass_alive), hash_(kZob.empty_board) {
  gid_.fill(-1);
  next_.fill(-1);
  libs_.fill(0);
  SeenInsert(hash_);  // board.cc:487-489: the empty position counts as seen
}

Board::Board(int handicap, float komi) : Board(komi, true) {
  if (handicap < 2 || handicap > 4) return;
  static constexpr Loc kStones[4] = {{15, 3}, {3, 15}, {3, 3}, {15, 15}};
  for (int i = 0; i < handicap; ++i) {
    int idx = Idx(kStones[i]);
    AddStone(idx, kBlack);
    hash_ ^= kZob.k[idx][kEmpty + 1] ^ kZob.k[idx][kBlack + 1];
  }
  local_n_ = 0;
  base_.reset();
  SeenInsert(hash_);
}

bool SeenTable::Contains(uint64_t h) const {
  for (uint32_t s = (uint32_t)mix(h) & (kCap - 1);; s = (s + 1) & (kCap - 1)) {
    if (slot[s] == h) return true;
    if (slot[s] == 0) return false;
  }
}

void SeenTable::Insert(uint64_t h) {
  if (count >= kCap / 2) return;  // unreachable within kMaxGameLen + read depth
  for (uint32_t s = (uint32_t)mix(h) & (kCap - 1);; s = (s + 1) & (kCap - 1)) {
    if (slot[s] == h) return;
    if (slot[s] == 0) {
      slot[s] = h;
      ++count;
      return;
    }
  }
}

bool Board::SeenContains(uint64_t h) const {
  if (h == 0) h = 1;
#ifdef __AVX2__
  // every legality check of the search and of the ladder reader ends here: four keys per compare (the list holds
  // up to kLocalCap = 96; the scalar early-exit loop was 8 % of the self-play host's time)
  {
    const __m256i key = _mm256_set1_epi64x((long long)h);
    int i = 0;
    for (; i + 4 <= local_n_; i += 4)
      if (_mm256_movemask_epi8(_mm256_cmpeq_epi64(_mm256_loadu_si256((const __m256i*)(local_ + i)), key))) return true;
    for (; i < local_n_; ++i)
      if (local_[i] == h) return true;
  }
#else
  for (int i = local_n_ - 1; i >= 0; --i)
    if (local_[i] == h) return true;
#endif
  return base_ && base_->Contains(h);
}

void Board::SeenInsert(uint64_t h) {
  if (h == 0) h = 1;
  if (local_n_ == kLocalCap) {  // flush the inline list into a fresh shared table
    auto t = std::make_shared<SeenTable>();
    if (base_) *t = *base_;
    for (int i = 0; i < local_n_; ++i) t->Insert(local_[i]);
    base_ = std::move(t);
    local_n_ = 0;
  }
  local_[local_n_++] = h;
}

int Board::EmptyNeighbors(int idx) const {  // GroupTracker::LibertiesAt, board.cc:143-152
  int l = 0;
  for (int k = 0; k < kNbr.n[idx]; ++k) l += stones_[kNbr.p[idx][k]] == kEmpty;
  return l;
}

// Removes a captured group: every vacated point becomes one new liberty of each distinct
// neighbouring group (GroupTracker::RemoveCaptures, board.cc:174-196).
void Board::RemoveGroup(int head) {
  int s = head;
  do {  // first pass: clear stones so that the group does not count as its own neighbour
    int nx = next_[s];
    hash_ ^= kZob.k[s][stones_[s] + 1] ^ kZob.k[s][kEmpty + 1];
    stones_[s] = kEmpty;
    s = nx;
  } while (s != head);
  s = head;
  do {
    int nx = next_[s];
    gid_[s] = -1;
    int seen[4], ns = 0;
    for (int k = 0; k < kNbr.n[s]; ++k) {
      int q = kNbr.p[s][k];
      if (stones_[q] == kEmpty) continue;
      int g = gid_[q];
      bool dup = false;
      for (int t = 0; t < ns; ++t) dup |= seen[t] == g;
      if (!dup) {
        seen[ns++] = g;
        ++libs_[g];
      }
    }
    next_[s] = -1;
    s = nx;
  } while (s != head);
}

// Puts a stone on an empty point whose captures are already resolved
// (GroupTracker::Move / NewGroup / AddToGroup / CoalesceGroups, board.cc:58-141,198-221):
// one liberty less for each distinct adjacent enemy group; own groups merge and the merged
// group's liberties are recounted exactly.
void Board::AddStone(int idx, Color color) {
  stones_[idx] = color;
  int friends[4], nf = 0, enemies[4], ne = 0;
  for (int k = 0; k < kNbr.n[idx]; ++k) {
    int q = kNbr.p[idx][k];
    if (stones_[q] == kEmpty) continue;
    int g = gid_[q];
    int* arr = stones_[q] == color ? friends : enemies;
    int& cnt = stones_[q] == color ? nf : ne;
    bool dup = false;
    for (int t = 0; t < cnt; ++t) dup |= arr[t] == g;
    if (!dup) arr[cnt++] = g;
  }
  for (int t = 0; t < ne; ++t) --libs_[enemies[t]];
  if (nf == 0) {
    gid_[idx] = (int16_t)idx;
    next_[idx] = (int16_t)idx;
    libs_[idx] = (int16_t)EmptyNeighbors(idx);
    return;
  }
  const int head = friends[0];
  // splice the other groups and the new stone into head's circular list
  for (int t = 1; t < nf; ++t) {
    int h2 = friends[t];
    int s = h2;
    do {
      gid_[s] = (int16_t)head;
      s = next_[s];
    } while (s != h2);
    std::swap(next_[head], next_[h2]);
  }
  gid_[idx] = (int16_t)head;
  next_[idx] = next_[head];
  next_[head] = (int16_t)idx;
  if (nf == 1) {
    // One friendly group grows by a stone: the point stops being its liberty, and the stone's empty neighbours
    // that no other stone of the group touches are new ones — the same count the walk below gives, without
    // the walk (a quarter of the self-play host's time went into recounts and ladder liberty walks).
    int l = libs_[head] - 1;
    for (int k = 0; k < kNbr.n[idx]; ++k) {
      const int q = kNbr.p[idx][k];
      if (stones_[q] != kEmpty) continue;
      bool touched = false;
      for (int m = 0; m < kNbr.n[q]; ++m) {
        const int r = kNbr.p[q][m];
        touched |= r != idx && stones_[r] == color && gid_[r] == head;
      }
      l += !touched;
    }
    libs_[head] = (int16_t)l;
    return;
  }
  // groups merged: exact recount (their liberty sets may overlap)
  g_marks.next();
  int l = 0, s = head;
  do {
    for (int k = 0; k < kNbr.n[s]; ++k) {
      int q = kNbr.p[s][k];
      if (stones_[q] == kEmpty && !g_marks.test_and_set(q)) ++l;
    }
    s = next_[s];
  } while (s != head);
  libs_[head] = (int16_t)l;
}

void Board::PlaceRaw(Loc loc, Color color) {
  int idx = Idx(loc);
  if (stones_[idx] != kEmpty) return;
  AddStone(idx, color);
  hash_ ^= kZob.k[idx][kEmpty + 1] ^ kZob.k[idx][color + 1];
}

bool Board::IsValidMove(Loc loc, Color color) const {
  if (loc == kPassLoc) return true;
  return MoveOk(PlayMoveDry(loc, color));
}

MoveStatus Board::PlayMoveDry(Loc loc, Color color, uint64_t* new_hash) const {
  if (loc == kPassLoc) {
    if (new_hash) *new_hash = hash_;
    return MoveStatus::kValid;
  }
  if (color != kBlack && color != kWhite) return MoveStatus::kUnknownColor;
  if (loc.i < 0 || loc.i >= kBoardLen || loc.j < 0 || loc.j >= kBoardLen) return MoveStatus::kOutOfBounds;
  const int idx = Idx(loc);
  if (stones_[idx] != kEmpty) return MoveStatus::kLocNotEmpty;
  if (prohibit_pass_alive_ && pass_alive_[idx] != kEmpty) return MoveStatus::kPassAliveRegion;

  // adjacent enemy groups in atari are captured (GetCapturedGroups, board.cc:990-1004)
  int cap[4], nc = 0;
  bool friend_not_atari = false, has_empty = false;
  for (int k = 0; k < kNbr.n[idx]; ++k) {
    int q = kNbr.p[idx][k];
    if (stones_[q] == kEmpty) {
      has_empty = true;
    } else if (stones_[q] == color) {
      if (libs_[gid_[q]] != 1) friend_not_atari = true;
    } else if (libs_[gid_[q]] == 1) {
      int g = gid_[q];
      bool dup = false;
      for (int t = 0; t < nc; ++t) dup |= cap[t] == g;
      if (!dup) cap[nc++] = g;
    }
  }
  // IsSelfCapture, board.cc:901-915: no capture, every adjacent own group in atari, no
  // empty neighbour
  if (nc == 0 && !friend_not_atari && !has_empty) return MoveStatus::kSelfCapture;

  uint64_t h = hash_ ^ kZob.k[idx][kEmpty + 1] ^ kZob.k[idx][color + 1];
  for (int t = 0; t < nc; ++t) {
    int s = cap[t];
    do {
      h ^= kZob.k[s][stones_[s] + 1] ^ kZob.k[s][kEmpty + 1];
      s = next_[s];
    } while (s != cap[t]);
  }
  if (SeenContains(h)) return MoveStatus::kRepeatedPosition;  // positional superko
  if (new_hash) *new_hash = h;
  return MoveStatus::kValid;
}

MoveStatus Board::PlayMove(Loc loc, Color color) {
  if (loc == kPassLoc) return Pass(color);
  uint64_t h;
  MoveStatus st = PlayMoveDry(loc, color, &h);
  if (!MoveOk(st)) return st;
  const int idx = Idx(loc);
  for (int k = 0; k < kNbr.n[idx]; ++k) {
    int q = kNbr.p[idx][k];
    if (stones_[q] == Opp(color) && libs_[gid_[q]] == 1) {
      int head = gid_[q], n = 0, s = head;
      do { ++n; s = next_[s]; } while (s != head);
      (color == kBlack ? w_prisoners_ : b_prisoners_) += n;
      RemoveGroup(head);
    }
  }
  AddStone(idx, color);
  consecutive_passes_ = 0;
  ++move_count_;
  hash_ = h;
  SeenInsert(hash_);
  return MoveStatus::kValid;
}

MoveStatus Board::Pass(Color) {
  ++consecutive_passes_;
  ++passes_;
  if (!IsGameOver() && prohibit_pass_alive_ && passes_ >= kNumPassesBeforeBensons)
    CalculatePassAliveRegions();
  return MoveStatus::kValid;
}

bool Board::IsAllPassAlive() {
  CalculatePassAliveRegions();
  return std::all_of(pass_alive_.begin(), pass_alive_.end(), [](Color c) { return c != kEmpty; });
}

void Board::CalculatePassAliveRegions() {
  pass_alive_.fill(kEmpty);
  CalculatePassAliveRegionForColor(kBlack);
  CalculatePassAliveRegionForColor(kWhite);
}

// Benson's algorithm as the reference runs it (board.cc:246-441):
//  * regions = maximal connected sets of non-`color` points that contain an empty point,
//    found from each unseen EMPTY point; a region is "small" iff each of its empty points
//    touches a `color` stone;
//  * a small region is vital to a group iff every empty point of it is a liberty of it;
//  * repeatedly drop groups with < 2 vital regions together with every small region they
//    touch; what survives (groups + regions, enemy stones inside included) is pass-alive.
void Board::CalculatePassAliveRegionForColor(Color color) {
  struct Region {
    std::vector<int16_t> locs;
    std::vector<int16_t> vital;   // group heads
    bool alive = true;
  };
  std::vector<Region> regions;
  // groups of `color`
  int16_t group_slot[kNumLocs];  // head -> index into groups, -1 otherwise
  std::fill(group_slot, group_slot + kNumLocs, (int16_t)-1);
  struct Group {
    int16_t head;
    int num_vital = 0;
    bool alive = true;
    std::vector<int16_t> adj_regions;
  };
  std::vector<Group> groups;
  for (int p = 0; p < kNumLocs; ++p)
    if (stones_[p] == color && gid_[p] == p) {
      group_slot[p] = (int16_t)groups.size();
      groups.emplace_back();
      groups.back().head = (int16_t)p;
    }
  if (groups.empty()) return;

  bool seen[kNumLocs] = {};
  std::vector<int16_t> stack;
  for (int p0 = 0; p0 < kNumLocs; ++p0) {
    if (seen[p0]) continue;
    if (stones_[p0] != kEmpty) {
      if (stones_[p0] == color) seen[p0] = true;
      continue;  // enemy stones never start a region
    }
    g_marks.next();
    Region reg;
    bool small = true;
    stack.clear();
    stack.push_back((int16_t)p0);
    g_marks.test_and_set(p0);
    while (!stack.empty()) {
      int p = stack.back();
      stack.pop_back();
      reg.locs.push_back((int16_t)p);
      seen[p] = true;
      bool is_liberty = stones_[p] != kEmpty;  // enemy stones do not need to be liberties
      for (int k = 0; k < kNbr.n[p]; ++k) {
        int q = kNbr.p[p][k];
        if (stones_[q] == color) {
          if (stones_[p] == kEmpty) is_liberty = true;
          continue;
        }
        if (!g_marks.test_and_set(q)) stack.push_back((int16_t)q);
      }
      if (!is_liberty) small = false;
    }
    if (small) regions.push_back(std::move(reg));
  }

  for (size_t r = 0; r < regions.size(); ++r) {
    Region& reg = regions[r];
    // adjacency (PopulateAdjacentRegions, board.cc:343-358)
    for (int16_t p : reg.locs)
      for (int k = 0; k < kNbr.n[p]; ++k) {
        int q = kNbr.p[p][k];
        if (stones_[q] != color) continue;
        Group& g = groups[group_slot[gid_[q]]];
        if (g.adj_regions.empty() || g.adj_regions.back() != (int16_t)r) {
          if (std::find(g.adj_regions.begin(), g.adj_regions.end(), (int16_t)r) == g.adj_regions.end())
            g.adj_regions.push_back((int16_t)r);
        }
      }
    // vital groups = groups adjacent to EVERY empty point (PopulateVitalRegions, :360-404)
    bool first = true;
    for (int16_t p : reg.locs) {
      if (stones_[p] != kEmpty) continue;
      int16_t adj[4];
      int na = 0;
      for (int k = 0; k < kNbr.n[p]; ++k) {
        int q = kNbr.p[p][k];
        if (stones_[q] == color) adj[na++] = gid_[q];
      }
      if (first) {
        for (int t = 0; t < na; ++t)
          if (std::find(reg.vital.begin(), reg.vital.end(), adj[t]) == reg.vital.end()) reg.vital.push_back(adj[t]);
        first = false;
      } else {
        reg.vital.erase(std::remove_if(reg.vital.begin(), reg.vital.end(),
                                       [&](int16_t g) { return std::find(adj, adj + na, g) == adj + na; }),
                        reg.vital.end());
      }
    }
    for (int16_t g : reg.vital) ++groups[group_slot[g]].num_vital;
  }

  // RunBenson, board.cc:406-441
  for (;;) {
    std::vector<int> drop;
    for (size_t g = 0; g < groups.size(); ++g)
      if (groups[g].alive && groups[g].num_vital < 2) drop.push_back((int)g);
    if (drop.empty()) break;
    for (int g : drop) {
      for (int16_t r : groups[g].adj_regions) {
        if (!regions[r].alive) continue;
        for (int16_t v : regions[r].vital) --groups[group_slot[v]].num_vital;
        regions[r].alive = false;
      }
      groups[g].alive = false;
    }
  }

  for (const Group& g : groups) {
    if (!g.alive) continue;
    int s = g.head;
    do {
      pass_alive_[s] = color;
      s = next_[s];
    } while (s != g.head);
  }
  for (const Region& r : regions)
    if (r.alive)
      for (int16_t p : r.locs) pass_alive_[p] = color;
}

Scores Board::GetScores() {
  CalculatePassAliveRegions();
  Scores sc;
  Grid bo, wo;
  sc.black_score = ScoreAndOwnership(kBlack, bo);
  sc.white_score = ScoreAndOwnership(kWhite, wo);
  for (int p = 0; p < kNumLocs; ++p)
    sc.ownership[p] = bo[p] == kBlack ? kBlack : (wo[p] == kWhite ? kWhite : kEmpty);
  return sc;
}

// Area score of one colour (board.cc:917-988): own stones that are not dead (inside the
// opponent's pass-alive area) + every region of empty points and enemy stones, bounded by
// own stones, that touches an own stone and holds no living enemy stone (dead enemy stones
// inside count as territory).  White gets komi.
float Board::ScoreAndOwnership(Color color, Grid& ownership) const {
  bool counted[kNumLocs] = {};
  ownership.fill(kEmpty);
  int score = 0;
  std::vector<int16_t> stack, region;
  for (int p0 = 0; p0 < kNumLocs; ++p0) {
    if (counted[p0]) continue;
    if (stones_[p0] == color) {
      if (pass_alive_[p0] != Opp(color)) {
        ++score;
        ownership[p0] = color;
      }
      counted[p0] = true;
      continue;
    }
    if (stones_[p0] == Opp(color)) {
      counted[p0] = true;
      continue;
    }
    g_marks.next();
    stack.clear();
    region.clear();
    stack.push_back((int16_t)p0);
    g_marks.test_and_set(p0);
    int region_score = 0;
    bool seen_self = false, seen_opp = false;
    while (!stack.empty()) {
      int p = stack.back();
      stack.pop_back();
      if (stones_[p] == color) {
        seen_self = true;
        continue;
      }
      if (stones_[p] == Opp(color)) {
        if (pass_alive_[p] == color) {
          region.push_back((int16_t)p);
          ++region_score;
        } else {
          seen_opp = true;
        }
      } else {
        region.push_back((int16_t)p);
        ++region_score;
      }
      counted[p] = true;
      for (int k = 0; k < kNbr.n[p]; ++k) {
        int q = kNbr.p[p][k];
        if (!g_marks.test_and_set(q)) stack.push_back((int16_t)q);
      }
    }
    if (seen_self && !seen_opp) {
      score += region_score;
      for (int16_t p : region) ownership[p] = color;
    }
  }
  return (float)score + (color == kWhite ? komi_ : 0.0f);
}

Grid Board::GetStonesWithLiberties(int liberties) const {
  Grid data{};
  for (int p = 0; p < kNumLocs; ++p)
    if (stones_[p] != kEmpty && libs_[gid_[p]] == liberties) data[p] = stones_[p];
  return data;  // empty points never match (the reference's group_info_map_[-1] read, §9)
}

// ---------------------------------------------------------------------------------------
// Ladder reader (board.cc:692-899).  A group in atari is "laddered" when it cannot get out:
// the defender extends at its liberty (or captures an adjacent attacker group in atari),
// the attacker ataris again from either of the two liberties, to depth 300.  Every move goes
// through full PlayMove legality (suicide, superko, pass-alive prohibition).  All choices are
// any/all quantifiers over the candidate moves, so the verdict does not depend on visiting
// order and this restatement is free to enumerate liberties its own way.
std::atomic<long> g_ladder_budget{0};   // 0 = unbounded (reference-exact), see SetLadderNodeBudget
thread_local long t_ladder_nodes = 0;
thread_local long t_ladder_budget = 0;   // the process-wide budget, read once per call
thread_local bool t_ladder_exhausted = false;
std::atomic<long> g_ladder_calls{0}, g_ladder_nodes{0}, g_ladder_max{0}, g_ladder_exhausted{0};

struct LadderSolver {
  // The first (up to) two liberties of the group in walk order.  Callers read groups of one or two liberties and
  // use out[0] (and out[1]): the walk stops as soon as it has as many as the group's exact count says there are.
  static int FindLiberties(const Board& b, int head, int out[2]) {
    const int want = b.libs_[head] < 2 ? b.libs_[head] : 2;
    int n = 0, s = head;
    out[0] = out[1] = head;   // (defined for a group without liberties: never the case for the callers)
    if (want <= 0) return 0;
    do {
      for (int k = 0; k < kNbr.n[s]; ++k) {
        int q = kNbr.p[s][k];
        if (b.stones_[q] == kEmpty && (n == 0 || q != out[0])) {
          out[n++] = q;
          if (n == want) return n;
        }
      }
      s = b.next_[s];
    } while (s != head);
    return n;
  }

  // distinct adjacent enemy groups with exactly one liberty
  static int SurroundingInAtari(const Board& b, int head, int out[64]) {
    int n = 0, s = head;
    const Color c = b.stones_[head];
    do {
      for (int k = 0; k < kNbr.n[s]; ++k) {
        int q = kNbr.p[s][k];
        if (b.stones_[q] == Opp(c) && b.libs_[b.gid_[q]] == 1) {
          int g = b.gid_[q];
          bool dup = false;
          for (int t = 0; t < n; ++t) dup |= out[t] == g;
          if (!dup && n < 64) out[n++] = g;
        }
      }
      s = b.next_[s];
    } while (s != head);
    return n;
  }

  // `board` already holds the position BEFORE last_move; plays it for the side that just
  // moved (the opponent of color_to_move) and reads on.
  static bool Solve(Board& board, Color g_color, Color color_to_move, int root, int last_move, int depth) {
    if (depth > 300) return false;
    // Optional work bound (off by default; not in the reference, whose only bound is the depth):
    // the read-out is exponential on chaotic positions with many groups in atari, and in a
    // batched self-play host one stalled game stalls its whole evaluation batch.  An exhausted
    // budget reads as "not laddered", like the depth bound.
    ++t_ladder_nodes;
    if (t_ladder_budget > 0 && t_ladder_nodes > t_ladder_budget) { t_ladder_exhausted = true; return false; }
    if (!MoveOk(board.PlayMove(AsLoc(last_move), Opp(color_to_move)))) return g_color != color_to_move;
    const int gid = board.gid_[root];
    if (gid < 0) return true;  // captured (not reachable through the reads below)
    const int liberties = board.libs_[gid];
    auto continuation = [&](int l) {
      Board copy = board;
      return Solve(copy, g_color, Opp(color_to_move), root, l, depth + 1);
    };
    // the last continuation of a node reads on in `board` itself: nothing looks at it afterwards (the caller's
    // copy is dropped when this call returns), and a board copy per ply was a sixth of the host's time
    auto last_continuation = [&](int l) { return Solve(board, g_color, Opp(color_to_move), root, l, depth + 1); };
    if (g_color != color_to_move) {  // attacker to move
      if (liberties > 2) return false;
      if (liberties <= 1) return true;
      int l[2];
      FindLiberties(board, gid, l);
      return continuation(l[0]) || last_continuation(l[1]);
    }
    // defender to move
    if (liberties > 1) return false;
    int l[2];
    FindLiberties(board, gid, l);
    if (!continuation(l[0])) return false;   // (most nodes end here: the group walk below only after it)
    int atari[64];
    const int na = SurroundingInAtari(board, gid, atari);
    for (int t = 0; t < na; ++t) {
      int nl[2];
      FindLiberties(board, atari[t], nl);
      if (t + 1 == na) return last_continuation(nl[0]);
      if (!continuation(nl[0])) return false;
    }
    return true;
  }
};

Grid Board::GetLadderedStones() const {
  Grid data{};
  for (int p = 0; p < kNumLocs; ++p) {
    if (stones_[p] == kEmpty || gid_[p] != p || libs_[p] != 1) continue;  // each atari group once
    int l[2];
    LadderSolver::FindLiberties(*this, p, l);
    if (EmptyNeighbors(l[0]) >= 3) continue;  // IsLaddered pre-check, board.cc:857-861
    const Color g_color = stones_[p];
    Board copy = *this;
    t_ladder_nodes = 0;
    t_ladder_exhausted = false;
    t_ladder_budget = g_ladder_budget.load(std::memory_order_relaxed);
    // the reference anchors the group by its root stone; any stone of it works since the
    // group can only grow while it is being read
    const bool laddered = LadderSolver::Solve(copy, g_color, Opp(g_color), p, l[0], 0);
    g_ladder_calls.fetch_add(1, std::memory_order_relaxed);
    g_ladder_nodes.fetch_add(t_ladder_nodes, std::memory_order_relaxed);
    if (t_ladder_exhausted) g_ladder_exhausted.fetch_add(1, std::memory_order_relaxed);
    for (long m = g_ladder_max.load(std::memory_order_relaxed); t_ladder_nodes > m &&
         !g_ladder_max.compare_exchange_weak(m, t_ladder_nodes, std::memory_order_relaxed);) {}
    if (laddered) {
      int s = p;
      do {
        data[s] = g_color;
        s = next_[s];
      } while (s != p);
    }
  }
  return data;
}

void SetLadderNodeBudget(long nodes) { g_ladder_budget.store(nodes < 0 ? 0 : nodes, std::memory_order_relaxed); }
long LadderNodeBudget() { return g_ladder_budget.load(std::memory_order_relaxed); }

void LadderStats(long out[4]) {
  out[0] = g_ladder_calls.load(); out[1] = g_ladder_nodes.load(); out[2] = g_ladder_max.load(); out[3] = g_ladder_exhausted.load();
}

// ---------------------------------------------------------------------------------------
Game::Game(float komi, bool prohibit_pass_alive) : board_(komi, prohibit_pass_alive) {
  moves_.assign(kMoveOffset, Move{kEmpty, kNoopLoc});
}

Game::Game(const Board& board, const Move last_moves[5], int init_mv_num)
    : board_(board), moves_(last_moves, last_moves + kMoveOffset), init_mv_num_(init_mv_num) {}

bool Game::PlayMove(Loc loc, Color color) {
  bool ok = MoveOk(board_.PlayMove(loc, color));
  if (ok) moves_.push_back(Move{color, loc});
  return ok;
}

void Game::WriteResult() {
  Scores s = board_.GetScores();
  result_.winner = s.black_score > s.white_score ? kBlack : (s.white_score > s.black_score ? kWhite : kEmpty);
  result_.bscore = s.black_score;
  result_.wscore = s.white_score;
  result_.by_resign = false;
  result_.ownership = s.ownership;
}

}  // namespace p3
