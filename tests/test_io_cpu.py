"""Host-side result tables (scripts/run_mmctm.jl:184-245) on a stub model: formats and arithmetic, no GPU."""
import numpy as np

import mmm_pkg

mmm_pkg.load()
from multimodalmusig_jl_amd import io as mio  # noqa: E402


class Stub:
    M = 2; K = [2, 1]; V = [3, 2]; D = 2
    γ = [[np.array([1.0, 1.0, 2.0]), np.array([3.0, 1.0, 0.0])], [np.array([1.0, 3.0])]]

    def lam_matrix(self):
        return np.array([[0.0, np.log(3.0), 5.0], [1.0, 1.0, -2.0]])


def test_cov2cor():
    C = np.array([[4.0, 2.0], [2.0, 9.0]])
    np.testing.assert_allclose(mio.cov2cor(C), [[1.0, 1 / 3], [1 / 3, 1.0]])


def test_topic_and_props_tables(tmp_path):
    m = Stub()
    rows = mio.topic_table(m, [["a", "b", "c"], ["x", "y"]], ["snv", "sv"])
    assert len(rows) == 2 * 3 + 1 * 2
    assert rows[0] == ("snv", 1, 1, "a", 0.25) and rows[5] == ("snv", 2, 3, "c", 0.0) and rows[7] == ("sv", 1, 2, "y", 0.75)
    labels, P = mio.props_table(m, ["s1", "s2"], ["snv", "sv"])
    assert labels == ["snv-1", "snv-2", "sv-1"]
    np.testing.assert_allclose(P, [[0.25, 0.5], [0.75, 0.5], [1.0, 1.0]])
    f = tmp_path / "sigs.tsv"; mio.write_sigs(f, m, [["a", "b", "c"], ["x", "y"]], ["snv", "sv"])
    lines = f.read_text().splitlines()
    assert lines[0] == "modality\ttopic\tvalue\tterm\tprobability" and lines[1] == "snv\t1\t1\ta\t0.25" and len(lines) == 9
    f = tmp_path / "props.tsv"; mio.write_props(f, m, ["s1", "s2"], ["snv", "sv"])
    lines = f.read_text().splitlines()
    assert lines[0] == "topic\ts1\ts2" and lines[3].split("\t") == ["sv-1", "1.0", "1.0"]
    f = tmp_path / "mu.tsv"; mio.write_matrix(f, np.array([1.5, -2.0]))
    assert f.read_text() == "1.5\n-2.0\n"
    f = tmp_path / "S.tsv"; mio.write_matrix(f, np.eye(2))
    assert f.read_text() == "1.0\t0.0\n0.0\t1.0\n"
