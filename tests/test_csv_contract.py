"""The CSV our driver writes must satisfy the reference's consumer (/root/reference/plot_results.py:3-46).

tests/golden/csv_expected.json holds what the reference's own parse_results() returned (in the build
container, tests/golden/make_csv_expect.py) for (a) a benchmark_results.csv written by driver/fa_driver on
an MI355X and (b) a synthetic file with the edge cases the parser handles. `parse_contract` below is the
rule set SURVEY.md section 8 row a11 extracts from it; the test checks rule set == reference output, and
that the driver's header/columns are exactly the reference's (main.mm:604-605,873-876). CPU only."""
import json
import os

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HEADER = "N,Naive(ms),Flash(ms),FlashV2(ms),FlashV3(ms),FlashV4(ms),SpeedupV1,SpeedupV2,SpeedupV3,SpeedupV4"


def parse_contract(path):
    n, s1, s2, s3, s4 = [], [], [], [], []
    started = False
    for line in open(path):
        if "N,Naive(ms)" in line:  # plot_results.py:16-18
            started = True
            continue
        if not started:
            continue
        parts = line.strip().split(",")
        if not line.strip() or len(parts) < 8:  # plot_results.py:22
            continue
        try:  # only fields 0-5 are used (plot_results.py:24-32)
            nn = int(parts[0])
            naive, t1, t2, t3, t4 = (float(parts[i]) for i in range(1, 6))
        except ValueError:
            continue
        if naive > 0:  # plot_results.py:34: rows without a naive time are dropped
            n.append(nn)
            for lst, t in ((s1, t1), (s2, t2), (s3, t3), (s4, t4)):
                lst.append(naive / t if t > 0 else 0)
    return [n, s1, s2, s3, s4]


def test_contract_matches_reference_parser_outputs():
    exp = json.load(open(os.path.join(HERE, "csv_expected.json")))
    for name, want in exp.items():
        got = parse_contract(os.path.join(HERE, name))
        assert got == want, name
    syn = exp["csv_synthetic.csv"]
    assert syn[0] == [128, 256, 1024]  # short row, junk row, blank line and the naive=0 row are dropped
    assert syn[1] == [0.5, 4.0 / 3.0, 0]  # a zero time gives speedup 0, extra trailing columns are ignored


def test_driver_csv_schema_is_the_reference_schema():
    lines = open(os.path.join(HERE, "benchmark_results_mi355x.csv")).read().splitlines()
    assert lines[0] == HEADER  # main.mm:604-605
    rows = [l.split(",") for l in lines[1:] if l]
    assert [int(r[0]) for r in rows] == [128, 256, 512, 1024, 2048, 4096, 8192, 16384]  # main.mm:608
    assert all(len(r) == 10 for r in rows)
    last = rows[-1]
    assert float(last[1]) == 0 and [float(x) for x in last[6:]] == [0, 0, 0, 0]  # naive skipped above 8192 (main.mm:673,862)
    for r in rows[:-1]:  # speedup columns are naive/time (main.mm:862-865)
        for j in range(4):
            assert abs(float(r[6 + j]) - float(r[1]) / float(r[2 + j])) < 2e-3 * float(r[6 + j])
    got = parse_contract(os.path.join(HERE, "benchmark_results_mi355x.csv"))
    assert got[0] == [128, 256, 512, 1024, 2048, 4096, 8192]  # the plotter keeps the seven rows with a naive time


def test_reference_parser_itself_when_present():
    ref_py = "/root/reference/plot_results.py"
    if not os.path.exists(ref_py):
        import pytest

        pytest.skip("reference not present (GPU box)")
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_plot", ref_py)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    for name in ("benchmark_results_mi355x.csv", "csv_synthetic.csv"):
        p = os.path.join(HERE, name)
        assert [list(x) for x in ref.parse_results(p)] == parse_contract(p)
