"""WatermarkID — 16-bit watermark identities.

Behavioural restatement of /root/reference/waveverify/watermark_id.py:16-376 (pure Python in the
reference too): the same factories, bit packings, hash choices (md5 first two bytes), license
code table, error types and equality/hash semantics, so existing callers keep working.
"""
from __future__ import annotations

import hashlib
from datetime import datetime
from typing import Any, Dict, Optional, Union

_LICENSES = {                       # watermark_id.py:159-169
    "CC0": 0x0000, "CC-BY": 0x0001, "CC-BY-SA": 0x0002, "CC-BY-NC": 0x0003,
    "CC-BY-NC-SA": 0x0004, "CC-BY-ND": 0x0005, "CC-BY-NC-ND": 0x0006,
    "ALL-RIGHTS": 0xFFFF, "CUSTOM": 0x8000,
}


def _md5_16(text: str) -> str:
    h = hashlib.md5(text.encode("utf-8")).digest()
    return "".join(format(b, "08b") for b in h[:2])


class WatermarkID:
    """A 16-bit message plus metadata about where it came from."""

    def __init__(self, bits: str):
        if not isinstance(bits, str):
            raise TypeError(f"Bits must be string, got {type(bits)}")
        if len(bits) != 16:
            raise ValueError(f"Bits must be exactly 16 characters, got {len(bits)}")
        if not all(c in "01" for c in bits):
            raise ValueError(f"Bits must contain only 0 and 1, got: {bits}")
        self.bits = bits
        self.metadata: Dict[str, Any] = {}

    # ------------------------------------------------------------------ factories
    @classmethod
    def for_creator(cls, creator_id: str) -> "WatermarkID":
        if not creator_id or not isinstance(creator_id, str):
            raise ValueError("Creator ID must be a non-empty string")
        w = cls(_md5_16(creator_id))
        w.metadata = {"type": "creator", "id": creator_id, "hash_method": "md5_first_2_bytes"}
        return w

    @classmethod
    def for_timestamp(cls, timestamp: Optional[datetime] = None) -> "WatermarkID":
        """5 bits year-2024, 4 bits month, 5 bits day, 2 bits quarter of day."""
        if timestamp is None:
            timestamp = datetime.now()
        off = timestamp.year - 2024
        if off < 0 or off > 31:
            raise ValueError(f"Year must be between 2024 and 2055, got {timestamp.year}")
        quarter = timestamp.hour // 6
        w = cls(f"{off:05b}{timestamp.month:04b}{timestamp.day:05b}{quarter:02b}")
        w.metadata = {"type": "timestamp", "time": timestamp.isoformat(), "year": timestamp.year,
                      "month": timestamp.month, "day": timestamp.day, "quarter": quarter}
        return w

    @classmethod
    def for_license(cls, license_type: str) -> "WatermarkID":
        norm = license_type.upper().replace("_", "-")
        if norm in _LICENSES:
            code = _LICENSES[norm]
        else:
            base = norm.split("-")[0] if "-" in norm else norm
            if base == "CC" and "-" in norm:
                parts = norm.split("-")
                base = "-".join(parts[: min(3, len(parts))])
            code = _LICENSES.get(base, _LICENSES["CUSTOM"])
        if code == _LICENSES["CUSTOM"]:
            h = hashlib.md5(license_type.encode()).digest()
            code = 0x8000 | (int.from_bytes(h[:2], "big") & 0x7FFF)
        w = cls(format(code, "016b"))
        w.metadata = {"type": "license", "license": license_type, "code": f"0x{code:04X}",
                      "is_custom": code >= 0x8000}
        return w

    @classmethod
    def for_tracking(cls, tracking_id: str) -> "WatermarkID":
        if not tracking_id or not isinstance(tracking_id, str):
            raise ValueError("Tracking ID must be a non-empty string")
        if tracking_id.isdigit() and len(tracking_id) <= 5 and int(tracking_id) <= 65535:
            bits, kind = format(int(tracking_id), "016b"), "numeric"
        else:
            bits, kind = _md5_16(tracking_id), "hashed"
        w = cls(bits)
        w.metadata = {"type": "tracking", "id": tracking_id, "id_type": kind}
        return w

    @classmethod
    def custom(cls, value: Union[str, int, bytes]) -> "WatermarkID":
        if isinstance(value, str):
            if len(value) == 16 and all(c in "01" for c in value):
                bits = value
            else:
                raise ValueError(f"String must be 16-bit binary (got {len(value)} chars). "
                                 f"Example: '1010101010101010'")
        elif isinstance(value, int):
            if 0 <= value <= 65535:
                bits = format(value, "016b")
            else:
                raise ValueError(f"Integer must be 0-65535, got {value}")
        elif isinstance(value, bytes):
            if len(value) == 2:
                bits = "".join(format(b, "08b") for b in value)
            else:
                raise ValueError(f"Bytes must be exactly 2 bytes, got {len(value)}")
        else:
            raise TypeError(f"Unsupported type {type(value)}. Use string, int, or bytes.")
        w = cls(bits)
        w.metadata = {"type": "custom", "value": str(value), "value_type": type(value).__name__}
        return w

    # ------------------------------------------------------------------ views
    def to_bits(self) -> str:
        return self.bits

    def to_hex(self) -> str:
        return format(int(self.bits, 2), "04X")

    def to_int(self) -> int:
        return int(self.bits, 2)

    def to_bytes(self) -> bytes:
        v = self.to_int()
        return bytes([(v >> 8) & 0xFF, v & 0xFF])

    def __str__(self) -> str:
        t = self.metadata.get("type", "unknown")
        if t == "creator":
            return f"WatermarkID(creator='{self.metadata['id']}')"
        if t == "timestamp":
            return f"WatermarkID(time='{self.metadata['time']}')"
        if t == "license":
            return f"WatermarkID(license='{self.metadata['license']}')"
        if t == "tracking":
            return f"WatermarkID(tracking='{self.metadata['id']}')"
        if t == "custom":
            return f"WatermarkID(custom={self.to_hex()})"
        return f"WatermarkID(bits='{self.bits}')"

    def __repr__(self) -> str:
        return f"WatermarkID(bits='{self.bits}', metadata={self.metadata})"

    def __eq__(self, other) -> bool:
        return isinstance(other, WatermarkID) and self.bits == other.bits

    def __hash__(self) -> int:
        return hash(self.bits)
