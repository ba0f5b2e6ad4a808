import json,sys
j=json.load(open(sys.argv[1]))
print(j["value"], j["ms_per_step"], j["config"]["check_ssim"][:30], j["config"]["cpu_affinity"])
for k in ("with_bitstream","single_stream","config3_literal","config5_literal","ref_shard"):
    v=j.get(k); print(k, {a:v[a] for a in v if a in ("value","ms_per_frame","fps","seconds","error","key_frames","bytes_gathered")} if v else None)
for k,v in j.get("other_configs",{}).items(): print(k, v["value"], v["ms_per_frame"])
print(j["cpu_baseline"]["value"] if "cpu_baseline" in j else None)
