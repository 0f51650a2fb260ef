import json,sys
for f in sys.argv[1:]:
    try:
        b=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, "value %.0f gn %.0f it/s (%.4f ms) frame %.3f ms" % (b["value"], b["gn"]["gn_iters_per_s"], b["gn"]["ms_per_gn_iter"], b["frame"]["ms_per_frame"]), {k:round(v,3) for k,v in b["frame"]["stage_ms_with_syncs"].items()}, "cost", b["gn"]["final_cost"], b["gn"].get("final_cost_rel_diff_vs_oracle"))
    except Exception as e:
        print(f, "ERR", e)
