"""Noise floor of the train-step comparison (tests/train_step_proxy.py) on a GPU box: how differently do two legs that run THE SAME code
(the reference's, over the shims) classify Gaussians at a densification event, run after run?  Both legs then differ only by the summation
order of the blending backward's float atomics (amplified by Adam over the segment).  Also re-runs the round-5 configuration (clone / split
boundary at the median splat size) to look for the clone <-> split tie behind GPUTEST_r05.  Writes one JSON (test infrastructure).

    python tools/train_step_noise_floor.py out.json [runs]
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import train_step_proxy as T  # noqa: E402


def summarise(rep):
    out = []
    for ev in rep["events"]:
        out.append({k: ev.get(k) for k in ("iteration", "kind", "accum_rel_l2", "opacity_rel_l2", "plan_A", "plan_B", "order_equal", "n_differently_classified",
                                           "differently_classified", "n_differently_pruned", "differently_pruned", "pruned", "xyz_rel_l2_after", "N_after")})
    return out


def main():
    out, runs = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
    res = {"reference_vs_reference": [], "dropins_vs_reference": [], "round5_config_median_boundary": []}
    for r in range(runs):
        res["reference_vs_reference"].append(summarise(T.run(iters=200, interval=50, leg_b="reference")))
        res["dropins_vs_reference"].append(summarise(T.run(iters=200, interval=50)))
        res["round5_config_median_boundary"].append(summarise(T.run(iters=200, interval=50, size_quantile=0.5)))
        json.dump(res, open(out, "w"), indent=1)
    worst = {}
    for k, rr in res.items():
        m = [max(abs(d["grad_margin_A"]) if (d["grad_margin_A"] >= 0) != (d["grad_margin_B"] >= 0) else 0.0,
                 abs(d["size_margin_A"]) if (d["size_margin_A"] > 0) != (d["size_margin_B"] > 0) else 0.0)
             for rep in rr for ev in rep for d in (ev.get("differently_classified") or [])]
        n_ev = sum(1 for rep in rr for ev in rep if ev.get("kind", "").startswith("densify"))
        worst[k] = {"densify_events": n_ev, "events_with_a_differently_classified_gaussian": sum(1 for rep in rr for ev in rep if ev.get("n_differently_classified")),
                    "largest_margin_of_a_flipped_decision": max(m) if m else 0.0,
                    "clone_split_flips": sum(1 for rep in rr for ev in rep for d in (ev.get("differently_classified") or []) if {d["A"], d["B"]} == {"clone", "split"})}
    res["summary"] = worst
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(worst, indent=1))


if __name__ == "__main__":
    main()
