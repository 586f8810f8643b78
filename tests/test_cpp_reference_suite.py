"""
Runs tests/cpp/reference_suite (built by __graft_entry__.build()): the reference crate's own unit tests restated in C++
against include/stark_anat.hpp — the compiled-language host side over the C ABI.  `host` mode needs no GPU (scalar field
ops, ProofStream KATs of proof_stream.rs:89-225); `gpu` mode adds the batched field ops, the Merkle suite of
merkle.rs:176-282 and the config-1 FRI commit / prove / verify pinned to the committed golden digests.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "reference_suite")


def _run(args):
    assert os.path.exists(BIN), f"{BIN} missing: run python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([BIN] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "test result: ok" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]
    return r.stdout


def test_reference_suite_host():
    out = _run(["host"])
    for name in ("field::test_mul", "field::test_div", "field::test_inv_minus_11", "proof_stream::test_prover_fiat_shamir",
                 "proof_stream::test_verifier_fiat_shamir", "proof_stream::test_serialize_inner"):
        assert f"test {name} ... ok" in out


@pytest.mark.gpu
def test_reference_suite_gpu(golden):
    out = _run(["gpu", golden("fri_c1.json")["stream_sha256"], golden("fri_prove_c1.json")["stream_sha256"]])
    for name in ("merkle::verify_path_for_each_leaf", "merkle::with_tampered_path", "merkle::different_tree_original_leaves_and_paths",
                 "fri::c1_commit_transcript_matches_golden", "fri::c1_prove_verify_matches_golden", "field::batched_ops_match_operators (GPU)"):
        assert f"test {name} ... ok" in out
