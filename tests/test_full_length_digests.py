"""The committed digests of the full-length oracle runs (tests/golden/full_length, made by scripts/full_length_oracle.py --oracle)
stay pinned to the oracle: the small sequence is regenerated in full, of every long one the first frames.  The GPU half --
the product against these digests, 300 frames of 4K and of 1080p -- is scripts/full_length_oracle.py --verify (log under
profiles/) and bench.py's self-checks."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("full_length_oracle", os.path.join(ROOT, "scripts", "full_length_oracle.py"))
flo = importlib.util.module_from_spec(spec)
spec.loader.exec_module(flo)


def test_the_small_sequence_regenerates_to_its_committed_digest():
    doc = flo.load("selftest")
    assert doc is not None, "tests/golden/full_length/selftest.json is part of the repository"
    c = flo.SEQUENCES["selftest"]
    W, H, rows = flo.oracle_run(c["W0"], c["H0"], c["seed"], c["frames"], c["gop"], c["start"])
    assert [W, H] == doc["coded"] and doc["key"] == [r[0] for r in rows] and sum(doc["key"]) == 2
    assert doc["frame_crc32"] == [r[1] for r in rows] and doc["frame_len"] == [r[2] for r in rows]
    assert doc["recon_crc32"] == [r[3] for r in rows]


@pytest.mark.parametrize("name", [n for n in flo.SEQUENCES if n != "selftest"] + list(flo.TABLES))
def test_the_head_of_every_long_digest_regenerates(name):
    doc = flo.load(name)
    if doc is None:
        pytest.skip(f"tests/golden/full_length/{name}.json not committed")
    if name in flo.SEQUENCES:
        c = flo.SEQUENCES[name]
        assert doc["frames"] == c["frames"] == len(doc["frame_crc32"]) == len(doc["recon_crc32"]) and doc["source"] == [c["W0"], c["H0"]]
        _, _, rows = flo.oracle_run(c["W0"], c["H0"], c["seed"], 2, c["gop"], c["start"])
        assert [r[1] for r in rows] == doc["frame_crc32"][:2] and [r[2] for r in rows] == doc["frame_len"][:2]
        assert [r[3] for r in rows] == doc["recon_crc32"][:2]
    else:
        c = flo.TABLES[name]
        # (the committed table may be shorter than what the script would generate today while a longer one is being made: its own length counts)
        phases = list(doc.get("phases", range(flo.ND)))
        assert len(doc["recon_crc32"]) == flo.ND and all(len(doc["recon_crc32"][ph]) == doc["frames"] for ph in phases) and 40 <= doc["frames"] <= c["frames"]
        assert float(doc.get("ssim_target", -1.0)) == float(c.get("ssim_target", -1.0)) and int(doc.get("conformant", 0)) == int(c.get("conformant", 0))
        ph = 3 if 3 in phases else phases[0]
        _, _, rows = flo.oracle_run(c["W0"], c["H0"], c["seed"], 2, 1 << 30, ph, refs=c.get("refs", "all"), want_bytes=False,
                                    ssim_target=c.get("ssim_target", -1.0), conformant=c.get("conformant", 0))
        assert [r[3] for r in rows] == doc["recon_crc32"][ph][:2]
