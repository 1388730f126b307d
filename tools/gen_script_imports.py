"""tests/golden/script_imports.json: which attributes the reference's sampling scripts take from the modules INTEGRATION.md A swaps
(read from the scripts' text in this container; the fixture is the list of names, not the scripts)."""
import json
import os
import re

REF = "/root/reference/scripts"
MODULES = ["dist_util", "inference_util", "test_util"]
out = {}
for script in ["video_sample.py", "video_sample_full.py", "video_nll.py"]:
    text = open(os.path.join(REF, script)).read()
    used = {m: sorted(set(re.findall(rf"\b{m}\.([A-Za-z_][A-Za-z0-9_]*)", text))) for m in MODULES}
    block = re.search(r"from improved_diffusion\.script_util import \(([^)]*)\)", text)
    used["script_util"] = sorted(n.strip() for n in block.group(1).replace("\n", " ").split(",") if n.strip()) if block else []
    out[script] = {m: v for m, v in used.items() if v}
json.dump(out, open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "script_imports.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
