"""Adaptive effect scheduler of the training step -- host logic, no device work.

Behavioural mirror of /root/reference/utils/effect_scheduler.py:39-808 (EffectScheduler): same class and method
names, arguments, exceptions, bookkeeping attributes and -- because the training loop seeds numpy's global
generator -- the same random-number calls in the same order, so a seeded run selects the same effects with the
same parameters.  Quirks kept on purpose: `select_effects` caps the count at the number of
known effects (watermarking.py:537 passes the batch size, so at most that many samples get effects), and nothing
in the training loop calls `adapt_effect_probabilities`, so probabilities stay uniform unless the caller adapts."""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

logger = logging.getLogger(__name__)


class EffectSchedulerError(Exception):
    """Base class of the scheduler's errors (effect_scheduler.py:16)."""


class InvalidEffectError(EffectSchedulerError):
    pass


class InvalidMetricError(EffectSchedulerError):
    pass


class ParameterValidationError(EffectSchedulerError):
    pass


def _ema(old: Optional[float], new: float, beta: float) -> float:
    return new if old is None else beta * old + (1 - beta) * new


class EffectScheduler:
    def __init__(self, effect_params: Dict[str, Dict[str, Any]], beta: float = 0.9, ber_threshold: float = 0.001,
                 miou_threshold: float = 0.95) -> None:
        if not 0 < beta < 1:
            raise ValueError(f"Beta must be in range (0, 1), got {beta}")
        if not 0 <= ber_threshold <= 1:
            raise ValueError(f"BER threshold must be in range [0, 1], got {ber_threshold}")
        if not 0 <= miou_threshold <= 1:
            raise ValueError(f"mIoU threshold must be in range [0, 1], got {miou_threshold}")
        try:
            self._validate_effect_params(effect_params)
        except Exception as e:
            raise ParameterValidationError(f"Invalid effect parameters: {str(e)}")
        names = list(effect_params.keys())
        self.effect_params = effect_params
        self.beta, self.ber_threshold, self.miou_threshold = beta, ber_threshold, miou_threshold
        self.effect_probabilities: Dict[str, float] = {n: 1.0 / len(names) for n in names}
        self.effect_usage_stats: Dict[str, int] = {n: 0 for n in names}
        self.total_effects = 0
        self.effect_metrics_history: Dict[str, Dict[str, Optional[float]]] = {n: {"ber": None, "miou": None} for n in names}
        self.current_effect_name: Optional[str] = None
        self.parameter_success_rates: Dict[str, Dict[Tuple[str, Any], List[bool]]] = {}
        self.effect_list: List[str] = names
        self.effect_ptr = 0
        self.parameter_metrics_history: Dict[str, Dict[Any, Dict[str, Any]]] = {n: {} for n in names}
        self.metric_history: Dict[str, Dict[str, Any]] = {
            n: {"overall": {"ber": [], "miou": []}, "params": {}} for n in names}

    # ---- selection (effect_scheduler.py:143-246) ------------------------------------------------------------------
    def _take(self, name: str) -> Tuple[str, Dict[str, Any]]:
        self.current_effect_name = name
        params = self._resolve_effect_params(self.effect_params.get(name, {}))
        self.effect_usage_stats[name] += 1
        self.total_effects += 1
        return name, params

    def select_all_effects(self) -> List[Tuple[str, Dict[str, Any]]]:
        try:
            return [self._take(n) for n in self.effect_params.keys()]
        except Exception as e:
            raise EffectSchedulerError(f"Effect selection failed: {str(e)}")

    def select_effects(self, num_effects: int = 3) -> List[Tuple[str, Dict[str, Any]]]:
        if num_effects <= 0:
            raise ValueError(f"Number of effects must be positive, got {num_effects}")
        try:
            names = list(self.effect_probabilities.keys())
            p = [self.effect_probabilities[n] for n in names]
            total = sum(p)
            p = [v / total for v in p] if total > 0 else [1.0 / len(names) for _ in names]
            drawn = np.random.choice(names, size=min(num_effects, len(names)), replace=True, p=p)
            return [self._take(n) for n in drawn]
        except Exception as e:
            raise EffectSchedulerError(f"Effect selection failed: {str(e)}")

    # ---- metrics (:252-430) ---------------------------------------------------------------------------------------
    def get_effect_probabilities(self) -> Dict[str, float]:
        return self.effect_probabilities.copy()

    def get_effect_statistics(self) -> Dict[str, Dict[str, Optional[float]]]:
        out: Dict[str, Dict[str, Optional[float]]] = {}
        try:
            for n in self.effect_params.keys():
                hist = self.metric_history[n]["overall"]
                out[n] = {
                    "usage_percentage": (self.effect_usage_stats[n] / self.total_effects * 100) if self.total_effects > 0 else 0.0,
                    "ema_ber": self.effect_metrics_history[n]["ber"],
                    "ema_miou": self.effect_metrics_history[n]["miou"],
                    "avg_ber": np.mean(hist["ber"]) if hist["ber"] else None,
                    "avg_miou": np.mean(hist["miou"]) if hist["miou"] else None,
                    "selection_count": self.effect_usage_stats[n],
                }
            return out
        except Exception:
            return {}

    def update_effect_metrics(self, effect_name: str, effect_params: Dict[str, Any], localized_ber: float, miou: float) -> None:
        if effect_name not in self.effect_params:
            raise InvalidEffectError(f"Unknown effect: '{effect_name}'")
        if not 0 <= localized_ber <= 1:
            raise InvalidMetricError(f"BER must be in range [0, 1], got {localized_ber}")
        if not 0 <= miou <= 1:
            raise InvalidMetricError(f"mIoU must be in range [0, 1], got {miou}")
        beta = self.beta
        ema = self.effect_metrics_history.setdefault(effect_name, {"ber": None, "miou": None})
        ema["ber"] = _ema(ema["ber"], localized_ber, beta)
        ema["miou"] = _ema(ema["miou"], miou, beta)
        hist = self.metric_history[effect_name]
        hist["overall"]["ber"].append(localized_ber)
        hist["overall"]["miou"].append(miou)
        key = self.make_hashable(effect_params)
        per = hist["params"].setdefault(key, {"ber": [], "miou": []})
        per["ber"].append(localized_ber)
        per["miou"].append(miou)
        success = localized_ber <= self.ber_threshold and miou >= self.miou_threshold
        rates = self.parameter_success_rates.setdefault(effect_name, {})
        for pname, pvalue in effect_params.items():
            rates.setdefault((pname, self.make_hashable(pvalue)), []).append(success)
        pm = self.parameter_metrics_history[effect_name].setdefault(key, {"ber": None, "miou": None, "count": 0})
        if pm["ber"] is None:
            pm["ber"], pm["miou"] = localized_ber, miou
        else:
            pm["ber"] = beta * pm["ber"] + (1 - beta) * localized_ber
            pm["miou"] = beta * pm["miou"] + (1 - beta) * miou
        pm["count"] += 1

    def adapt_effect_probabilities(self) -> None:
        """Reward 0.8 (1 - BER) + 0.2 mIoU per parameter set, averaged per effect, softmax, then 0.8 / 0.2
        smoothing against the old probabilities (:432-504)."""
        try:
            scores: Dict[str, float] = {}
            for n, per in self.parameter_metrics_history.items():
                r = [0.8 * (1 - m["ber"]) + 0.2 * m["miou"] for m in per.values()
                     if m["ber"] is not None and m["miou"] is not None]
                scores[n] = np.mean(r) if r else 0.0
            names = list(scores.keys())
            s = np.array([scores[n] for n in names])
            if np.all(s == 0):
                new = np.ones_like(s) / len(s)
            else:
                e = np.exp((s - np.max(s)) / 1.0)
                new = e / np.sum(e)
            for n, q in zip(names, new):
                self.effect_probabilities[n] = 0.8 * self.effect_probabilities[n] + (1 - 0.8) * q
            self._normalize_probabilities()
        except Exception as e:
            raise EffectSchedulerError(f"Probability adaptation failed: {str(e)}")

    def log_adaptive_behavior(self, logger_func: Optional[Any] = None) -> None:
        out = print if logger_func is None else logger_func
        try:
            out("\n" + "=" * 60)
            out("EFFECT SCHEDULER ADAPTIVE BEHAVIOR")
            out("=" * 60)
            out("\nEffect Selection Probabilities:")
            for n, p in sorted(self.effect_probabilities.items(), key=lambda kv: kv[1], reverse=True):
                out(f"  {n}: {p:.4f}")
            out("\nEffect Performance Statistics:")
            for n, st in sorted(self.get_effect_statistics().items()):
                out(f"\n  {n}:")
                out(f"    Usage: {st['usage_percentage']:.1f}%")
                for label, k in (("EMA BER", "ema_ber"), ("EMA mIoU", "ema_miou"), ("Avg BER", "avg_ber"), ("Avg mIoU", "avg_miou")):
                    if st[k] is not None:
                        out(f"    {label}: {st[k]:.4f}")
            out("=" * 60 + "\n")
        except Exception as e:
            logger.error(f"Failed to log adaptive behavior: {str(e)}")

    # ---- parameters (:560-747) ------------------------------------------------------------------------------------
    def _validate_effect_params(self, effect_params: Dict[str, Dict[str, Any]]) -> None:
        try:
            bp = effect_params.get("bandpass_filter") if "bandpass_filter" in effect_params else None
            if bp is not None and "cutoff_freq_low" in bp and "cutoff_freq_high" in bp:
                lows = bp.get("cutoff_freq_low", {}).get("choices", [])
                highs = bp.get("cutoff_freq_high", {}).get("choices", [])
                if lows and highs and not any(lo < hi for lo in lows for hi in highs):
                    raise ParameterValidationError(
                        f"Bandpass filter has no valid frequency combinations. Low frequencies {lows} must have at "
                        f"least one value less than high frequencies {highs}")
        except ParameterValidationError:
            raise
        except Exception as e:
            raise ParameterValidationError(f"Failed to validate parameters: {str(e)}")

    def _resolve_effect_params(self, raw_params: Dict[str, Any]) -> Dict[str, Any]:
        """A value per parameter: entries with 'choices' are drawn with weight (success rate + 0.1), unexplored
        values counting as 0.5 (:613-687)."""
        out: Dict[str, Any] = {}
        try:
            for key, cfg in raw_params.items():
                if not (isinstance(cfg, dict) and "choices" in cfg):
                    out[key] = cfg
                    continue
                choices = cfg["choices"]
                if not choices:
                    continue
                seen = self.parameter_success_rates.get(self.current_effect_name, {})
                w = []
                for ch in choices:
                    h = seen.get((key, self.make_hashable(ch)), [])
                    w.append((sum(h) / len(h) if h else 0.5) + 0.1)
                total = sum(w)
                if total > 0:
                    idx = np.random.choice(len(choices), p=[v / total for v in w])
                else:
                    idx = np.random.randint(len(choices))
                out[key] = choices[idx]
            if self.current_effect_name == "bandpass_filter":
                self._validate_bandpass_frequencies(out)
            return out
        except Exception as e:
            raise ParameterValidationError(f"Parameter resolution failed: {str(e)}")

    def _validate_bandpass_frequencies(self, params: Dict[str, Any]) -> None:
        lo, hi = params.get("cutoff_freq_low"), params.get("cutoff_freq_high")
        if lo is None or hi is None or lo < hi:
            return
        cfg = self.effect_params.get("bandpass_filter", {})
        highs = cfg.get("cutoff_freq_high", {}).get("choices", [])
        ok_hi = [f for f in highs if f > lo]
        if ok_hi:
            hi = np.random.choice(ok_hi)
        else:
            lows = cfg.get("cutoff_freq_low", {}).get("choices", [])
            ok_lo = [f for f in lows if f < hi]
            if ok_lo:
                lo = np.random.choice(ok_lo)
            else:
                lo = min(lows) if lows else lo
                hi = max(highs) if highs else hi
        params["cutoff_freq_low"], params["cutoff_freq_high"] = lo, hi

    # ---- utilities (:749-806) -------------------------------------------------------------------------------------
    def _normalize_probabilities(self) -> None:
        try:
            total = sum(self.effect_probabilities.values())
            if abs(total - 1.0) > 1e-6:
                total = max(total, 1e-10)
                for k in self.effect_probabilities:
                    self.effect_probabilities[k] /= total
            if abs(sum(self.effect_probabilities.values()) - 1.0) > 1e-6:
                for k in self.effect_probabilities:
                    self.effect_probabilities[k] = 1.0 / len(self.effect_probabilities)
        except Exception as e:
            raise EffectSchedulerError(f"Probability normalization failed: {str(e)}")

    def make_hashable(self, value: Any) -> Any:
        if isinstance(value, (list, tuple)):
            return tuple(self.make_hashable(v) for v in value)
        if isinstance(value, dict):
            return tuple(sorted((k, self.make_hashable(v)) for k, v in value.items()))
        if isinstance(value, np.ndarray):
            return tuple(value.tolist())
        return value
