"""CPU: the RULES that turn A/B runs into the two generated tables (tools/make_dispatch.py, tools/make_window_table.py), on synthetic
runs whose right answer is obvious -- the committed tables are checked against the committed runs elsewhere
(test_dispatch_table.py, test_window_table.py); this pins what the rules mean."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_dispatch_rule_one_block_forms():
    md = _load("make_dispatch")
    # (cell-column, [wave-per-block at 3 4 5 6 8 waves]) per box
    assert md.decide([(6000, [5000, 6500, 6400, 6300, 6200]), (6000, [5100, 6450, 6500, 6300, 6100])], 0.02) == 4      # best geometric mean
    assert md.decide([(7000, [5000, 6500, 6400, 6300, 6200]), (7000, [5100, 6450, 6500, 6300, 6100])], 0.02) == 0      # cell-column leads by > 2 % on EVERY box
    assert md.decide([(7000, [5000, 6500, 6400, 6300, 6200]), (6500, [5100, 6450, 6500, 6300, 6100])], 0.02) == 4      # ... not on the second one: a tie goes to the runtime-width kernel
    assert md.decide([(0, [1, 2, 3, 4, 5])], 0.02) == 3                                                              # no cell-column measurement


def test_dispatch_rule_two_blocks_per_wavefront():
    md = _load("make_dispatch")
    one = [(6000, [6100, 6300, 6300, 6250, 6200]), (6000, [6050, 6320, 6280, 6260, 6190])]
    two_better = [(6000, [6700, 6500, 6550, 6600, 6650]), (6000, [6800, 6480, 6560, 6610, 6640])]
    # the two-block form wins by > 2 % -- but never at 3 workgroups per CU (index 0), however good it looks there
    assert md.decide_two(one, two_better, 0.02) == md.TWO_BLOCKS + 8
    two_same = [(6000, [6700, 6350, 6340, 6300, 6250]), (6000, [6800, 6330, 6300, 6290, 6240])]
    assert md.decide_two(one, two_same, 0.02) == 4                         # within the margin (3 waves excluded): the one-block form stays
    cc_best = [(7000, w) for _, w in one]
    assert md.decide_two(cc_best, [(7000, w) for _, w in two_same], 0.02) == 0


def _window_file(tmp_path, name, rows):
    p = tmp_path / name
    lines = ["# synthetic", "case                                                                 w=0          w=31          w=16"]
    for op, ty, whole, windowed in rows:
        lines.append(f"{op} {ty} W=7 (1000 blocks)   {whole:6.0f} ({whole / 8000:.3f})   {whole:6.0f} ({whole / 8000:.3f})   {windowed:6.0f} ({windowed / 8000:.3f})")
    p.write_text("\n".join(lines) + "\n")
    return str(p)


def test_window_rule(tmp_path):
    mw = _load("make_window_table")
    a = _window_file(tmp_path, "a.txt", [("pack", "u32", 6000, 6400), ("unpack", "u32", 6800, 6500), ("delta", "u16", 6500, 6600), ("transpose", "u8", 6000, 6300),
                                         ("block_min_max", "u64", 6000, 6150)])
    b = _window_file(tmp_path, "b.txt", [("pack", "u32", 6100, 6350), ("unpack", "u32", 6700, 6600), ("delta", "u16", 6500, 6400), ("block_min_max", "u64", 6000, 6090)])
    table, why = mw.decide([a, b], 0.01)
    assert table[("pack", "u32")] == 16                 # never lost, median gain > 2 %
    assert table[("unpack", "u32")] == 31               # loses everywhere
    assert table[("delta", "u16")] == 31                # +1.5 % on one box, -1.5 % on the other: it lost by the margin somewhere
    assert table[("transpose", "u8")] == 31             # one run is not enough, however large the gain
    assert table[("block_min_max", "u64")] == 31        # never lost, but a median gain of 2.0 % is not MORE than twice the margin
    assert table[("undelta", "u64")] == 31 and why[("undelta", "u64")] == "-"     # nobody measured it
    text = mw.render([a, b], 0.01, table, why)
    assert "constexpr unsigned char PACK[4] = {31, 31, 16, 31};" in text
