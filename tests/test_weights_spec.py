"""Checkpoint layout: param_spec() must name exactly the tensors the reference eval path reads."""
import json
import os
import re

from isr2_amd.weights import param_spec, synth_state_dict, count_params

HERE = os.path.dirname(os.path.abspath(__file__))
DEAD = ("collaborative.", "freq_router.", "expert_weights", "band_importance")
ALIAS = ("expert_ensemble.nafnet.intro.", "expert_ensemble.nafnet.ending.", "expert_ensemble.nafnet.middle_blks.",
         "expert_ensemble.nafnet.body.")
BUFFERS = ("num_batches_tracked", "relative_position_index", "rpe_biases", "attn_mask_", "dct_basis", "_mask",
           "lo_row", "hi_row", "lo_col", "hi_col", "gaussian.kernel")


def test_spec_matches_reference_manifest():
    man = json.load(open(os.path.join(HERE, "golden", "state_manifest.json")))
    spec = {n: tuple(s) for n, s, _ in param_spec()}
    for n, s in spec.items():
        assert n in man, n
        assert tuple(man[n][0]) == s, (n, man[n][0], s)
    for n in man:
        if n in spec:
            continue
        assert n.startswith(DEAD) or n.startswith(ALIAS) or any(b in n for b in BUFFERS), f"unaccounted reference tensor {n}"


def test_param_counts():
    # SURVEY.md section 6 [measured]: HAT 40 846 575, NAFNet 115 982 915; fusion live = 1 017 906 - 291 467 dead
    assert count_params(("hat",)) == 40846575
    assert count_params(("nafnet",)) == 115982915
    assert count_params(("dat",)) > 14_000_000


def test_synth_is_deterministic_and_nontrivial():
    a = synth_state_dict(1234, parts=("fusion",))
    b = synth_state_dict(1234, parts=("fusion",))
    c = synth_state_dict(1, parts=("fusion",))
    for k in a:
        assert (a[k] == b[k]).all()
    assert any((a[k] != c[k]).any() for k in a)
    assert a["cross_band_attn.lka_block.norm1.running_var"].min() > 0.4
