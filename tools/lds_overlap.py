"""LDS overlap probe (dd_tools_lds_overlap_probe): workgroups of two concurrent dispatches with different dynamic-LDS sizes on the same CUs, each
holding and verifying a pattern in ALL of its LDS.   python tools/lds_overlap.py  -> one JSON line per (lds_a, lds_b) pair."""
import json, os, sys
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import _lib

torch.cuda.set_device(0)
L = _lib.load_tools()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
pairs = [(131072, 30720), (65536, 30720), (147456, 13312), (131072, 12288), (131072, 28160), (131072, 32768), (66560, 30720), (131840, 30720),
         (0, 30720), (0, 76800), (78080, 78080), (131072, 6656)]
if len(sys.argv) > 1:
    pairs = [tuple(int(x) for x in p.split(",")) for p in sys.argv[1:]]
out = []
for lds_a, lds_b in pairs:
    err = torch.tensor([0, 0, 0xFFFFFFFF - 2 ** 32, 0], dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for rep in range(6):
        rc = L.dd_tools_lds_overlap_probe(lds_a, 256 if lds_a else 0, 400, lds_b, 2048, 6, 12, err.data_ptr(), sa.cuda_stream, sb.cuda_stream)
        assert rc == 0, L.dd_last_error()
        torch.cuda.synchronize()
    e = [int(x) & 0xFFFFFFFF for x in err.tolist()]
    rec = {"lds_a_bytes": lds_a, "lds_b_bytes": lds_b, "corrupted_words_seen_by_a": e[0], "corrupted_words_seen_by_b": e[1],
           "b_first_bad_byte": None if e[1] == 0 else e[2] * 4, "b_last_bad_byte": None if e[1] == 0 else e[3] * 4 + 3}
    out.append(rec)
    print(json.dumps(rec), flush=True)
if os.environ.get("DD_OVERLAP_LOG"):
    json.dump(out, open(os.environ["DD_OVERLAP_LOG"], "w"), indent=1)
