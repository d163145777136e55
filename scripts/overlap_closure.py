"""Round 6: where the ideal of a perfectly overlapped job goes.  ONE table: the encoder alone, the decode loop alone, and
both at once (the product's pipelined schedule: encoder of the next batch on a budget of CUs beside the decode loop), each running for
several seconds while a sampler thread reads what the board says about itself -- power, shader clock, memory clock -- and the work done
is counted from the outside: TFLOP/s of the encoder (2.272 TFLOP per clip), TB/s of the decode loop's cross-K/V stream
(245.76 MB per utterance and token).

    python scripts/overlap_closure.py [batch] [tokens] [seconds per phase] > gpurun_out/r6_overlap_closure.txt

Reads /sys/class/drm/card*/device (hwmon power1_average | power1_input, pp_dpm_sclk, pp_dpm_mclk) as an ordinary user; falls back to
`rocm-smi --showpower --showclocks --json` when sysfs is not readable.  What the table cannot show (in-kernel clock under MFMA load reads up to
10 % below pp_dpm_sclk: MI355X_MICROARCH.md, DVFS give-back) is said in its footer."""
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import native  # noqa: E402,F401
import torch  # noqa: E402
import synthetic  # noqa: E402
from pathlib import Path  # noqa: E402
from decoding import WhisperDecoding  # noqa: E402
from encoding import WhisperEncoding  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 576
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
SECS = float(sys.argv[3]) if len(sys.argv) > 3 else 12.0
ENC_CUS = int(os.environ.get("ENC_CUS", "96"))


# ---- what the board says about itself ------------------------------------------------------------------------------------------------
def _first(pattern):
    for f in sorted(glob.glob(pattern)):
        try:
            open(f).read()
            return f
        except OSError:
            continue
    return None


CARD = None
for c in sorted(glob.glob("/sys/class/drm/card*/device")):
    if os.path.exists(os.path.join(c, "pp_dpm_sclk")):
        CARD = c
        break
POWER_F = _first(f"{CARD}/hwmon/hwmon*/power1_average") or _first(f"{CARD}/hwmon/hwmon*/power1_input") if CARD else None
SCLK_F = os.path.join(CARD, "pp_dpm_sclk") if CARD else None
MCLK_F = os.path.join(CARD, "pp_dpm_mclk") if CARD else None
TEMP_F = _first(f"{CARD}/hwmon/hwmon*/temp1_input") if CARD else None
CAP_F = _first(f"{CARD}/hwmon/hwmon*/power1_cap") if CARD else None


def _dpm_current(path):
    try:
        for line in open(path):
            if "*" in line:
                return float(line.split(":")[1].strip().split("M")[0].strip())
    except (OSError, ValueError, IndexError):
        pass
    return None


def _smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        card = d.get("card0") or next(iter(d.values()))
        p = next((float(v) for k, v in card.items() if "Power" in k and "W" in k), None)
        sclk = next((float(v.strip("()Mhz")) for k, v in card.items() if k.startswith("sclk")), None)
        mclk = next((float(v.strip("()Mhz")) for k, v in card.items() if k.startswith("mclk")), None)
        return p, sclk, mclk, None
    except Exception:       # noqa: BLE001
        return None, None, None, None


def sample():
    if POWER_F or SCLK_F:
        p = None
        try:
            p = float(open(POWER_F).read()) / 1e6 if POWER_F else None
        except (OSError, ValueError):
            pass
        t = None
        try:
            t = float(open(TEMP_F).read()) / 1e3 if TEMP_F else None
        except (OSError, ValueError):
            pass
        return p, _dpm_current(SCLK_F), _dpm_current(MCLK_F), t
    return _smi_sample()


class Sampler:
    def __init__(self, period=0.05):
        self.period, self.rows, self.stop = period, [], False
        self.th = threading.Thread(target=self.run, daemon=True)

    def run(self):
        while not self.stop:
            self.rows.append(sample())
            time.sleep(self.period)

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.th.join()

    def mean(self, i):
        v = [r[i] for r in self.rows if r[i] is not None]
        return sum(v) / len(v) if v else None

    def peak(self, i):
        v = [r[i] for r in self.rows if r[i] is not None]
        return max(v) if v else None


# ---- the two halves of the job ------------------------------------------------------------------------------------------------------------
eng = Path("/tmp/wm_bench_engines/large-v2-int8-seed0")
if not (eng / "decoder_config.json").exists():
    import argparse
    import bench
    eng.parent.mkdir(parents=True, exist_ok=True)
    bench.build_engines(argparse.Namespace(model="large-v2", config="int8", seed=0), eng)
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
dec.sample_len = T
g = torch.Generator(device="cuda").manual_seed(1234)
mel = (torch.randn((B, 80, 3000), generator=g, device="cuda") * 0.5).clamp_(-0.5, 1.5).half()
xa = enc.get_audio_features_async(mel)
dec.detect_language(xa)
dec.main_loop(xa, ignore_eot=True)          # captures the graphs
torch.cuda.synchronize()
FLOP_CLIP = 2.272e12
KV_BYTES = 245.76e6


def run_phase(name, body):
    """`body()` issues one unit of work and returns (encoder clips, decode token steps) it contained; repeated for SECS seconds."""
    torch.cuda.synchronize()
    clips = steps = 0
    with Sampler() as s:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < SECS:
            c, st = body()
            clips += c
            steps += st
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    return dict(name=name, wall=wall, clips=clips, steps=steps, tflops=clips * FLOP_CLIP / wall / 1e12, tbs=steps * B * KV_BYTES / wall / 1e12,
                power=s.mean(0), power_peak=s.peak(0), sclk=s.mean(1), mclk=s.mean(2), temp=s.mean(3), n=len(s.rows))


def enc_alone():
    enc.get_audio_features_async(mel)
    torch.cuda.synchronize()
    return B, 0


def dec_alone():
    dec.main_loop(xa, ignore_eot=True)
    torch.cuda.synchronize()
    return 0, T


def both():
    # the product's pipelined schedule (bench.py's step): the next batch's encoder on ENC_CUS CUs beside this batch's decode loop; the
    # budget is given back when the loop ends, the rest of the pass takes the whole chip
    enc.prefetch(mel, ENC_CUS)
    dec.main_loop(xa, ignore_eot=True)
    enc.loop_ended()
    enc.collect()
    torch.cuda.synchronize()
    return B, T


rows = [run_phase("idle", lambda: (time.sleep(0.2), (0, 0))[1])] if SECS >= 2 else []
rows += [run_phase("encoder alone (whole chip)", enc_alone), run_phase("decode loop alone", dec_alone),
         run_phase(f"both: encoder on {ENC_CUS} CUs beside the loop", both)]
cap_w = None
try:
    cap_w = float(open(CAP_F).read()) / 1e6 if CAP_F else None
except (OSError, ValueError):
    pass
print(f"# overlap closure, large-v2 int8, B = {B}, T = {T}, {SECS:.0f} s per phase; sensors: power {POWER_F}, sclk {SCLK_F}, mclk {MCLK_F}, temp {TEMP_F}; "
      f"board power cap (power1_cap): {cap_w} W")
print("| phase | wall s | encoder passes | token steps | encoder TFLOP/s | cross-K/V TB/s | board W (mean / peak) | sclk MHz | mclk MHz | temp C | samples |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
f = lambda v, d=1: "--" if v is None else f"{v:.{d}f}"      # noqa: E731
for r in rows:
    print(f"| {r['name']} | {r['wall']:.2f} | {r['clips'] / B:.1f} | {r['steps']} | {f(r['tflops'])} | {f(r['tbs'], 2)} | {f(r['power'], 0)} / {f(r['power_peak'], 0)} | "
          f"{f(r['sclk'], 0)} | {f(r['mclk'], 0)} | {f(r['temp'])} | {r['n']} |")
ea, da, bo = rows[-3], rows[-2], rows[-1]
t_enc, t_dec = ea["wall"] / (ea["clips"] / B), da["wall"] / (da["steps"] / T)
t_both = bo["wall"] / (bo["clips"] / B)
print(f"\nper batch: encoder alone {t_enc * 1e3:.0f} ms, decode loop alone {t_dec * 1e3:.0f} ms -> one after the other {1e3 * (t_enc + t_dec):.0f} ms, "
      f"perfect overlap max() = {1e3 * max(t_enc, t_dec):.0f} ms, measured together {t_both * 1e3:.0f} ms "
      f"({(t_enc + t_dec) / t_both:.3f} x the sequential rate; the ideal would be {(t_enc + t_dec) / max(t_enc, t_dec):.3f} x)")
print(f"shares while together: the encoder ran at {bo['tflops'] / ea['tflops']:.2f} of its alone rate, the K/V stream at {bo['tbs'] / da['tbs']:.2f} of its alone rate "
      f"(sum {bo['tflops'] / ea['tflops'] + bo['tbs'] / da['tbs']:.2f}: 1.0 = zero-sum, 2.0 = free overlap)")
print("note: pp_dpm_sclk is the DPM state's nominal clock; under MFMA load the in-kernel clock reads up to 10 % below it (MI355X_MICROARCH.md, DVFS give-back 6) -- "
      "the power column is the witness that the encoder half runs at the board's limit")
