"""Transcribes the reference's own rules known-answer tests into data fixtures.

    python tests/golden/make_rules_fixtures.py     (build container only; needs /root/reference)

Reads cc/game/__tests__/board_test.cc and cc/game/__tests__/symmetry_test.cc as TEXT and
emits tests/golden/board_cases.json / symmetry_cases.json: for every SUBCASE the sequence of
board operations and the expected observations (move legality, stones, liberty planes,
pass-alive regions, scores/ownership, ladder verdicts).  Only data is kept: positions,
moves and expected values; no reference code.  GroupTracker-level subcases (NewGroup /
AddToGroup with no capture logic) become raw stone placements, group ids become the
location that created them.
"""
import json
import os
import re
import sys

REF = "/root/reference/cc/game/__tests__"
OUT = os.path.dirname(os.path.abspath(__file__))
COL = {"BLACK": 1, "WHITE": -1, "EMPTY": 0}
LOC = r"(?:game::)?Loc\{\s*(-?\d+)\s*,\s*(-?\d+)\s*\}"


def locs_in(s):
    return [[int(a), int(b)] for a, b in re.findall(LOC, s)]


def split_subcases(text):
    out = []
    for m in re.finditer(r'SUBCASE\("([^"]+)"\)\s*\{', text):
        i = m.end()
        depth = 1
        while depth:
            c = text[i]
            if c == '"' and text[i - 1] == 'R':  # raw string
                j = text.index(')"', i)
                i = j + 2
                continue
            depth += c == "{"
            depth -= c == "}"
            i += 1
        out.append((m.group(1), text[m.end():i - 1]))
    return out


def statements(body):
    # protect raw strings, strip comments, split on ';'
    raws = []

    def keep(m):
        raws.append(m.group(1))
        return f"__RAW{len(raws) - 1}__"
    body = re.sub(r'R"\((.*?)\)"', keep, body, flags=re.S)
    body = re.sub(r"//[^\n]*", "", body)
    stmts = [re.sub(r"\s+", " ", s).strip() for s in body.split(";")]
    return [s for s in stmts if s], raws


def translate(name, body):
    stmts, raws = statements(body)
    ops, var_loc, var_col, unknown = [], {}, {}, []
    cur_plane = {}
    for s in stmts:
        m = re.fullmatch(r"(?:game::)?(?:Board|GroupTracker) \w+", s)
        if m or s in ("using game::BoardToDSL", "using game::ParseBoardDSL", "using game::ParseBoardGrid"):
            continue
        m = re.fullmatch(r"(?:auto|game::Board|Board) (\w+) = (?:game::)?ParseBoardDSL\(__RAW(\d+)__\)", s)
        if m:
            ops.append(["dsl", raws[int(m.group(2))]])
            continue
        if re.fullmatch(r"auto board_copy = board", s) or "steady_clock" in s or "duration_cast" in s or \
                s.startswith("MESSAGE(") or s == ".count()" or s == "CHECK_EQ(board, board_copy)":
            continue
        m = re.fullmatch(r"board\.Play(Black|White)\((\d+), (\d+)\)", s)
        if m:
            ops.append(["play", COL[m.group(1).upper()], int(m.group(2)), int(m.group(3))])
            continue
        m = re.fullmatch(r"(CHECK|CHECK_FALSE)\(MoveOk\(board\.Play(Black|White)\((\d+), (\d+)\)\)\)", s)
        if m:
            ops.append(["play_expect", COL[m.group(2).upper()], int(m.group(3)), int(m.group(4)), m.group(1) == "CHECK"])
            continue
        m = re.fullmatch(r"(CHECK|CHECK_FALSE)\(MoveOk\(board\.PlayMoveDry\(" + LOC + r", (\w+)\)\)\)", s)
        if m:
            ops.append(["dry_expect", COL[m.group(4)], int(m.group(2)), int(m.group(3)), m.group(1) == "CHECK"])
            continue
        m = re.fullmatch(r"(CHECK|CHECK_FALSE)\(MoveOk\(board\.PlayMove\(" + LOC + r", (\w+)\)\)\)", s)
        if m:
            ops.append(["play_expect", COL[m.group(4)], int(m.group(2)), int(m.group(3)), m.group(1) == "CHECK"])
            continue
        m = re.fullmatch(r"board\.PlayMove\(" + LOC + r", (\w+)\)", s)
        if m:
            ops.append(["play", COL[m.group(3)], int(m.group(1)), int(m.group(2))])
            continue
        m = re.fullmatch(r"board\.Pass\((\w+)\)", s)
        if m:
            ops.append(["pass", COL[m.group(1)]])
            continue
        m = re.fullmatch(r"CHECK\(board\.at\((\d+), (\d+)\) == (\w+)\)", s) or \
            re.fullmatch(r"CHECK_EQ\(board\.at\((\d+), (\d+)\), (\w+)\)", s)
        if m:
            ops.append(["at", int(m.group(1)), int(m.group(2)), COL[m.group(3)]])
            continue
        m = re.fullmatch(r"(CHECK|CHECK_FALSE)\(board\.IsAllPassAlive\(\)\)", s)
        if m:
            ops.append(["all_pass_alive", m.group(1) == "CHECK"])
            continue
        # planes
        m = re.fullmatch(r"(?:Board::BoardData|auto) (\w+) = board\.GetStonesInAtari\(\)", s)
        if m:
            cur_plane[m.group(1)] = ["libs_plane", 1]
            continue
        m = re.fullmatch(r"(?:Board::BoardData|auto) (\w+) = board\.GetStonesWithLiberties\((\d)\)", s)
        if m:
            cur_plane[m.group(1)] = ["libs_plane", int(m.group(2))]
            continue
        m = re.fullmatch(r"(?:Board::BoardData|auto) (\w+) = board\.GetLadderedStones\(\)", s)
        if m:
            cur_plane[m.group(1)] = ["ladder_plane"]
            continue
        m = re.fullmatch(r"CHECK\((\w+)\[" + LOC + r"\] == (\w+)\)", s) or \
            re.fullmatch(r"CHECK\((\w+)\[AsIndex\(" + LOC + r", BOARD_LEN\)\] == (\w+)\)", s)
        if m and m.group(1) in cur_plane:
            ops.append(cur_plane[m.group(1)] + [int(m.group(2)), int(m.group(3)), COL[m.group(4)]])
            continue
        m = re.fullmatch(r"CHECK\((\w+)\[" + LOC + r"\] != EMPTY\)", s)
        if m and m.group(1) in cur_plane:
            ops.append([cur_plane[m.group(1)][0] + "_nonempty"] + cur_plane[m.group(1)][1:] + [int(m.group(2)), int(m.group(3))])
            continue
        if s == "bool all_pass_alive = board.IsAllPassAlive()":
            ops.append(["calc_all_pa"])
            continue
        m = re.fullmatch(r'const std::string s = ((?:"[XO.+]+" ?)+)', s)
        if m:
            cur_plane["__str"] = "".join(re.findall(r'"([XO.+]+)"', m.group(1)))
            continue
        if s == "auto board = ParseBoard(s)":
            ops.append(["parse_seq", cur_plane["__str"]])
            continue
        if name == "NewBoardIsEmpty":
            if ["all_empty"] not in ops:
                ops.append(["all_empty"])
            continue
        # scores
        if s == "Scores scores = board.GetScores()":
            ops.append(["get_scores"])
            continue
        m = re.fullmatch(r"CHECK\(scores\.(black|white)_score == ([\d.]+)\)", s)
        if m:
            ops.append(["score", m.group(1), float(m.group(2))])
            continue
        m = re.fullmatch(r"CHECK\(OwnershipRegionsMatch\( ?scores\.ownership, \{(.*?)\}, \{(.*?)\}\)\)", s)
        if m:
            ops.append(["ownership", locs_in(m.group(1)), locs_in(m.group(2))])
            continue
        # group tracker level
        m = re.fullmatch(r"(?:groupid (\w+) = )?group_tracker\.NewGroup\(" + LOC + r", (\w+)\)", s)
        if m:
            if m.group(1):
                var_loc[m.group(1)] = [int(m.group(2)), int(m.group(3))]
                var_col[m.group(1)] = COL[m.group(4)]
            ops.append(["raw", COL[m.group(4)], int(m.group(2)), int(m.group(3))])
            continue
        m = re.fullmatch(r"group_tracker\.AddToGroup\(" + LOC + r", (\w+)\)", s)
        if m:
            ops.append(["raw", var_col[m.group(3)], int(m.group(1)), int(m.group(2))])
            continue
        m = re.fullmatch(r"groupid (\w+) = group_tracker\.CoalesceGroups\(" + LOC + r"\)", s)
        if m:
            var_loc[m.group(1)] = [int(m.group(2)), int(m.group(3))]
            continue
        m = re.fullmatch(r"CHECK\(group_tracker\.LibertiesForGroup\((\w+)\) == (\d+)\)", s)
        if m:
            ops.append(["libs", var_loc[m.group(1)], int(m.group(2))])
            continue
        m = re.fullmatch(r"CHECK\(group_tracker\.LibertiesForGroupAt\(" + LOC + r"\) == (\d+)\)", s)
        if m:
            ops.append(["libs", [int(m.group(1)), int(m.group(2))], int(m.group(3))])
            continue
        m = re.fullmatch(r"CHECK\(group_tracker\.LibertiesForGroupAt\(" + LOC + r"\) == group_tracker\.LibertiesForGroupAt\(" + LOC + r"\)\)", s)
        if m:
            ops.append(["libs_eq", [int(m.group(1)), int(m.group(2))], [int(m.group(3)), int(m.group(4))]])
            continue
        m = re.fullmatch(r"CHECK\(group_tracker\.GroupAt\(" + LOC + r"\) == (\w+)\)", s)
        if m:
            ops.append(["same_group", [int(m.group(1)), int(m.group(2))], var_loc[m.group(3)]])
            continue
        m = re.fullmatch(r"group_tracker\.CalculatePassAliveRegionForColor\((\w+)\)", s)
        if m:
            ops.append(["calc_pa", COL[m.group(1)]])
            continue
        m = re.fullmatch(r"absl::flat_hash_set<Loc> (\w+) = \{(.*)\}", s)
        if m:
            cur_plane[m.group(1)] = [[int(a), int(b)] for a, b in re.findall(r"\{\s*(\d+)\s*,\s*(\d+)\s*\}", m.group(2))]
            continue
        m = re.fullmatch(r"CHECK\(PaRegionsMatch\(group_tracker, std::move\((\w+)\), (\w+)\)\)", s)
        if m:
            ops.append(["pa_region", COL[m.group(2)], cur_plane[m.group(1)]])
            continue
        unknown.append(s)
    return ops, unknown


def main():
    if not os.path.isdir(REF):
        print("reference not present; keeping committed fixtures")
        return
    text = open(os.path.join(REF, "board_test.cc")).read()
    cases, skipped = [], []
    for name, body in split_subcases(text):
        ops, unknown = translate(name, body)
        if unknown:
            skipped.append((name, unknown))
            continue
        cases.append({"name": name, "ops": ops})
    with open(os.path.join(OUT, "board_cases.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print(len(cases), "board cases;", len(skipped), "skipped")
    for n, u in skipped:
        print("  SKIP", n, "|", u[:3])


if __name__ == "__main__":
    main()


def symmetry_fixture():
    """The eight 5x5 golden grids and the eight Loc known answers of symmetry_test.cc."""
    text = open(os.path.join(REF, "symmetry_test.cc")).read()
    names = ["grid", "grid_rot90", "grid_rot180", "grid_rot270", "grid_flip", "grid_flip_rot90",
             "grid_flip_rot180", "grid_flip_rot270"]
    grids = []
    for n in names:
        m = re.search(r"kGridSize> " + n + r" = \{(.*?)\}", text, flags=re.S)
        grids.append([int(v) for v in re.findall(r"\d+", m.group(1))])
    syms = ["kIdentity", "kRot90", "kRot180", "kRot270", "kFlip", "kFlipRot90", "kFlipRot180", "kFlipRot270"]
    locs = []
    for sname in syms:
        m = re.search(r"CHECK_EQ\(ApplySymmetry\(Symmetry::" + sname + r", loc, grid_len\), Loc\{(\d+), (\d+)\}\)", text)
        locs.append([int(m.group(1)), int(m.group(2))])
    with open(os.path.join(OUT, "symmetry_cases.json"), "w") as f:
        json.dump({"grid_len": 5, "grids": grids, "loc_grid_len": 9, "loc": [2, 3], "loc_images": locs}, f)
    print("symmetry fixture ok")


if __name__ == "__main__" and os.path.isdir(REF):
    symmetry_fixture()
