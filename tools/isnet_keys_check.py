"""Diff the committed IS-Net key / shape list (tests/golden/isnet_dis_state_dict_keys.json, generated
from the restatement in drawingspinup_amd/mv/matting.py) against a real DIS checkpoint.

    python tools/isnet_keys_check.py /path/to/isnet-general-use.pth

Exit code 0: every entry of the list is in the checkpoint with the same shape (keys the checkpoint
has beyond the list — training-time modules — are printed, not an error)."""
import json
import os
import sys

import torch

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = json.load(open(os.path.join(root, "tests", "golden", "isnet_dis_state_dict_keys.json")))["entries"]
    sd = torch.load(sys.argv[1], map_location="cpu", weights_only=True)
    sd = sd.get("state_dict", sd)
    bad = [(k, s, tuple(sd[k].shape) if k in sd else None) for k, s in want.items()
           if k not in sd or list(sd[k].shape) != s]
    extra = sorted(set(sd) - set(want))
    print(f"{len(want) - len(bad)} / {len(want)} entries match; {len(extra)} checkpoint keys beyond the list")
    for b in bad[:40]:
        print("MISMATCH", b)
    for k in extra[:40]:
        print("extra", k, tuple(sd[k].shape))
    sys.exit(1 if bad else 0)
