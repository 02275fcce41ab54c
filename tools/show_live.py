import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d["tracking"]
print(t.get("live_single_frame")); print(t.get("live_single_frame_device_state")); print(d.get("errors"), t.get("error"))
