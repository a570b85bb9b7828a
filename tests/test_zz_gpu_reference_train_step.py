"""The reference's own train step (GaussianModel.training_setup / update_learning_rate / compute_regulation / add_densification_stats /
densify / prune / reset_opacity + render() + l1_loss + torch.optim.Adam, byte-compiled from /root/reference) executed on the MI355X over this
repository's `diff_gaussian_rasterization` / `simple_knn` shims, against the same 200 iterations through the drop-ins (fdgs.render,
fdgs.deform_network, fdgs.losses, fdgs.compute_regulation, fdgs.densify.*, fdgs.FusedAdam) from identical state: train.py:180-292.

(File name: collected LAST -- a multi-minute harness must not be able to hide the tests behind it from a `-x` run.)"""
import json
import os

import numpy as np
import pytest

import train_step_proxy

pytestmark = pytest.mark.gpu

# A Gaussian may be classified differently by the two legs only when it STRADDLES a threshold: the margins of the decision that fell
# differently have opposite signs in the two legs and the reference leg's margin is within TIE_FACTOR x the event's own `accum_rel_l2`
# (how far the statistic the decision reads differs between the legs over that segment -- itself bounded below), at least TIE_FLOOR.
# Measured (profiles/r06_train_step_noise_floor.json, 72 densification events on MI355X): two legs that BOTH run the reference's code differ
# by accum_rel_l2 1e-7 ... 5e-4 per segment (summation order of the blending backward's float atomics, amplified by Adam), and the reference
# leg's own plan differs between runs by up to 6 Gaussians at the fourth event; drop-ins vs reference: 1e-5 ... 1.1e-3, a flipped decision
# in 7 of 48 events, margins in the reference leg 3e-6 ... 1.3e-3 (at accum_rel_l2 6.6e-4), every one a straddle; one of them a clone <->
# split flip at size margins -4e-5 / +2e-4 -- N stays equal and every later row shifts by one: the signature of GPUTEST_r05's failure
# (`xyz_rel_l2_after = 1.288`), which the round-5 harness could not tell from a defect.
TIE_FACTOR, TIE_FLOOR = 3.0, 1e-4


def _skip_without_ref():
    from oracle import ref_modules
    if not ref_modules.available() or not os.path.isfile(os.path.join(ref_modules.OUT, "scene", "gaussian_model.pyc")):
        pytest.skip("oracle/_ref not built (python -m oracle.build_ref where /root/reference exists)")


def _show(rep):
    for ev in rep["events"]:
        print({k: (float(f"{v:.3e}") if isinstance(v, float) else v) for k, v in ev.items()})


def _check_straddles(ev, tie):
    for d in ev.get("differently_classified", []):
        # which of the two decisions fell differently, and how close to its threshold was that Gaussian in the reference leg?
        # (the decisions as the legs made them in float32 -- not the sign of a reported margin, which one ulp from zero may differ: run 2 of
        # gpurun_out/gpu_full_2.log, `split` at grad margin -6e-8)
        grad_flipped = d["grad_decision_A"] != d["grad_decision_B"]
        size_flipped = d["size_decision_A"] != d["size_decision_B"]
        assert grad_flipped or size_flipped, (d, "classified differently with BOTH decisions equal: a defect, not a tie")
        if grad_flipped:
            assert abs(d["grad_margin_A"]) < tie and abs(d["grad_margin_A"] - d["grad_margin_B"]) < 2 * tie, (d, tie, ev)
        if size_flipped:
            assert abs(d["size_margin_A"]) < tie and abs(d["size_margin_A"] - d["size_margin_B"]) < 2 * tie, (d, tie, ev)
    for d in ev.get("differently_pruned", []):
        flipped = [k for k in ("opacity", "screen", "world") if (d[f"{k}_margin_A"] > 0) != (d[f"{k}_margin_B"] > 0)]
        assert flipped and all(abs(d[f"{k}_margin_A"]) < max(tie, 3 * ev["opacity_rel_l2"]) for k in flipped if k != "screen"), (d, ev)
        # (the screen-size term compares integer radii: a flip there is a radius that differs by one pixel between the legs)
        assert all(abs(d["screen_margin_A"] - d["screen_margin_B"]) * ev["prune_screen_size"] <= 1.0 + 1e-6 for k in flipped if k == "screen"), (d, ev)


def _check_event(ev, xyz_after_tol=1e-4):
    """One densification / prune event of two legs that entered it with the same N."""
    assert ev["N_before"][0] == ev["N_before"][1]
    tie = max(TIE_FLOOR, TIE_FACTOR * ev["accum_rel_l2"])
    _check_straddles(ev, tie)
    if "prune" in ev["kind"]:
        assert ev["n_differently_pruned"] <= 2 and abs(ev["pruned"][0] - ev["pruned"][1]) <= ev["n_differently_pruned"], ev
        if ev["n_differently_pruned"] == 0:
            assert ev["pruned"][0] == ev["pruned"][1] and ev["xyz_rel_l2_after_prune"] < 1e-4, ev
    if "densify" not in ev["kind"]:
        return
    assert ev["n_differently_classified"] <= 3, ev
    if ev["n_differently_classified"] == 0:
        assert ev["order_equal"] and ev["plan_A"] == ev["plan_B"] and ev["N_after"][0] == ev["N_after"][1], ev
        assert ev["rows_only_in_one_leg"] == 0, ev
    else:
        assert ev["rows_only_in_one_leg"] <= 4 * ev["n_differently_classified"], ev
    # every row both legs hold (matched by parent + slot -- in the same order when nothing was classified differently) agrees to rounding
    assert ev["table_equal"] and all(v < 5e-3 for v in ev["matched_rows_rel_l2"].values()), ev
    assert ev["xyz_rel_l2_after"] < xyz_after_tol, ev


def test_reference_train_step_over_the_shims_matches_the_drop_ins():
    _skip_without_ref()
    rep = train_step_proxy.run(iters=200, interval=50)
    out = os.environ.get("FDGS_TRAIN_STEP_JSON")
    if out:
        json.dump(rep, open(out, "w"))
    _show(rep)
    ma, mb = float(np.mean(rep["final_psnr_A"])), float(np.mean(rep["final_psnr_B"]))
    dens = [ev for ev in rep["events"] if "densify" in ev["kind"]]
    print(f"densification threshold {rep['threshold']:.3e}; N {rep['events'][0]['N_before'][0]} -> {rep['N_final']}; resyncs {rep['resyncs']}; "
          f"PSNR start {rep['psnr_A'][0]:.2f} dB -> reference loop {ma:.3f} dB, drop-ins {mb:.3f} dB (diff {mb - ma:+.4f}); max per-iteration drift {rep['drift']:.4f} dB")
    assert rep["optimizer_B"] == "FusedAdam"
    assert len(dens) == 4 and rep["N_final"][0] > rep["events"][0]["N_before"][0]                      # the set actually grew
    assert any(ev["plan_B"][1] > 0 for ev in dens) and any(ev["plan_B"][2] > 0 for ev in dens)         # clones AND splits happened
    assert any(ev.get("pruned", (0, 0))[0] > 20 for ev in rep["events"])                               # and the prune dropped something
    for ev in rep["events"]:
        # the statistics the densification decides on, accumulated since the segment started from identical state (measured over 24 runs,
        # profiles/r06_train_step_noise_floor.json: accum <= 1.1e-3 -- the fourth segment, which holds the opacity reset --, denom identical,
        # max_radii2D differing on <= 0.09 % of the Gaussians by one pixel, positions to <= 2e-6)
        assert ev["accum_rel_l2"] < 5e-3 and ev["denom_mismatch_frac"] < 2e-3 and ev["max_radii2D_mismatch_frac"] < 5e-3 and ev["xyz_rel_l2"] < 1e-4, ev
        _check_event(ev)
    assert dens[0]["n_differently_classified"] == 0 or dens[0]["accum_rel_l2"] > 1e-5
    assert sum(ev["n_differently_classified"] == 0 for ev in dens) >= 2
    assert abs(mb - ma) < 0.05                                          # north_star: PSNR within 0.05 dB
    assert max(abs(a - b) for a, b in zip(rep["final_psnr_A"], rep["final_psnr_B"])) < 0.15
    assert np.abs(np.array(rep["psnr_A"][:40]) - np.array(rep["psnr_B"][:40])).max() < 0.01


def test_reference_train_step_free_running_legs():
    """The same two legs WITHOUT the per-event copy of leg A's state into leg B: the drop-ins' own trajectory through three densifications
    and a prune (their own restructured Parameters, Adam moments and statistics feed every later step -- a broken fdgs_densify_apply would
    show here).  The legs drift apart by rounding amplified through Adam, so later events may classify a Gaussian near a threshold
    differently (then, and only then, leg B is resynchronised): every such Gaussian is named and must straddle the threshold."""
    _skip_without_ref()
    rep = train_step_proxy.run(iters=150, interval=50, prune_at=80, reset_at=70, resync_every_event=False)
    _show(rep)
    ma, mb = float(np.mean(rep["final_psnr_A"])), float(np.mean(rep["final_psnr_B"]))
    print(f"free-running: N {rep['events'][0]['N_before'][0]} -> {rep['N_final']}; resyncs {rep['resyncs']}; PSNR reference loop {ma:.3f} dB, drop-ins {mb:.3f} dB")
    dens = [ev for ev in rep["events"] if "densify" in ev["kind"]]
    assert len(dens) == 3 and any(ev.get("pruned", (0, 0))[0] > 20 for ev in rep["events"])
    assert rep["resyncs"] == sum(1 for ev in rep["events"] if ev.get("n_differently_classified") or ev.get("n_differently_pruned"))
    for ev in rep["events"]:
        assert ev["accum_rel_l2"] < 2e-2 and ev["xyz_rel_l2"] < 1e-3, ev         # (free-running: the segment did not start from identical state)
        _check_event(ev, xyz_after_tol=1e-3)
    assert abs(mb - ma) < 0.05
