import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], "| device", d["device_resident"]["value"], "| abi", d["abi_path"].get("value"), d["abi_path"].get("same_strings_as_timed_path"), d["abi_path"].get("error"), "| f32", d["f32_strict"]["value"])
