import sys, json
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        a = d.get("advection", {})
        print(f.split("/")[-1], "value %.3e" % d["value"], "ms/step %.3f" % d["ms_per_step"], "cell_ms %.4f" % d["roofline"]["kernel_ms"],
              "frac %.3f" % d["roofline"]["frac"], "adv_ms %.3f" % a.get("ms_per_step", float("nan")), a.get("phase_ms"))
    except Exception as e:
        print(f, "ERR", e)
