"""Writes tests/golden/isnet_dis_state_dict_keys.json: every state_dict key and shape of the IS-Net
restatement (drawingspinup_amd/mv/matting.py ISNetDIS, after xuebinqin/DIS models/isnet.py).
The real checkpoint is not available here ("leaf unpinned"): tools/isnet_keys_check.py compares the
list with an isnet-general-use.pth on a machine that has one.

    python tests/golden/make_isnet_keys.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

if __name__ == "__main__":
    from drawingspinup_amd.mv import matting
    sd = matting.ISNetDIS().state_dict()
    path = os.path.join(HERE, "isnet_dis_state_dict_keys.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    out = {"note": old.get("note", ""), "entries": {k: list(v.shape) for k, v in sd.items()}}
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", len(out["entries"]), "entries")
