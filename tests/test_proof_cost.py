"""sipp_amd/proof_cost.py (the verifier-side price of level L-D): the word counts it derives from a proof's shape are
the lengths of real proofs (the committed digests of the oracle's n = 4 and n = 128 proofs carry them), and sharding an instance over
more ranks never makes the verifier's work smaller."""
import json

from sipp_amd import proof_cost


def test_proof_words_equal_the_committed_proofs():
    g4 = json.load(open("tests/golden/proof_digests_n4.json"))
    for kind, key, nio in ((0, "g1", 3), (1, "g2", 3), (2, "fq12", 4)):
        assert proof_cost.stark_verifier_cost(kind, nio)["proof_words"] == g4[key]["words"], key
    g = json.load(open("tests/golden/proof_digests_n128.json"))
    for kind, key, nio in ((0, "g1", 127), (1, "g2", 127), (2, "fq12", 14), (4, "g1_hardened", 127), (5, "g2_hardened", 127)):
        assert proof_cost.stark_verifier_cost(kind, nio)["proof_words"] == g[key]["words"], key


def test_price_of_io_sharding_grows_with_the_ranks():
    lists = [4095, 4095, 24]
    rows = [proof_cost.instance_price(lists, w) for w in (1, 2, 4, 8)]
    assert [r["proofs"] for r in rows] == [3, 6, 12, 24]
    for a, b in zip(rows, rows[1:]):
        assert b["proof_words"] > a["proof_words"] and b["verifier_hashes"] > a["verifier_hashes"] and b["query_openings"] > a["query_openings"]
    one = proof_cost.instance_price(lists, 1)
    assert one["query_openings"] == 84 * ((3 + 4) + (3 + 4) + (3 + 2))      # N = 2^21 folds four times, the Fq12 STARK (2^14) twice
    # world 8 leaves ranks without an Fq12 record when the list is short
    assert proof_cost.instance_price([7, 7, 6], 8)["proofs"] == 7 + 7 + 6
