#!/usr/bin/env python3
"""Known answers for WatermarkID from the reference's own pure-Python class
(/root/reference/waveverify/watermark_id.py, loaded by path; build container only)."""
import importlib.util, json, os
from datetime import datetime
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_wid", "/root/reference/waveverify/watermark_id.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
W = m.WatermarkID
out = {"creator": {}, "tracking": {}, "license": {}, "timestamp": {}, "custom": {}, "str": {}}
for s in ["beyonce_2024", "john_doe_music", "a", "Ünïcode-ß", "x" * 300]:
    out["creator"][s] = W.for_creator(s).bits
for s in ["0", "42", "65535", "65536", "99999", "123456", "CASE-2024-001", "podcast-ep-123", "00042"]:
    w = W.for_tracking(s); out["tracking"][s] = [w.bits, w.metadata["id_type"]]
for s in ["CC0", "cc-by", "CC_BY_SA", "CC-BY-4.0", "CC-BY-NC-ND-4.0", "ALL-RIGHTS", "CUSTOM", "MIT", "GPL-3.0", "cc-zero", "CC"]:
    w = W.for_license(s); out["license"][s] = [w.bits, w.metadata["code"], w.metadata["is_custom"]]
for iso in ["2024-01-01T00:00:00", "2025-07-17T12:34:56", "2055-12-31T23:59:59", "2031-02-28T06:00:00"]:
    w = W.for_timestamp(datetime.fromisoformat(iso)); out["timestamp"][iso] = w.bits
for v in [0, 42, 65535, "1010101010101010", b"\xab\xcd"]:
    w = W.custom(v); out["custom"][repr(v)] = [w.bits, w.to_hex(), w.to_int(), list(w.to_bytes()), str(w)]
out["str"]["creator"] = str(W.for_creator("abc")); out["str"]["license"] = str(W.for_license("MIT"))
out["str"]["tracking"] = str(W.for_tracking("77"))
json.dump(out, open(os.path.join(HERE, "watermark_ids.json"), "w"), indent=1, ensure_ascii=False)
print("ok")
