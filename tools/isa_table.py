"""Static per-function table of a gfx950 assembly file (hipcc -save-temps ... .s): code size, VGPRs, scratch size,
scratch loads / stores outside the prologue / epilogue (callee-saved register saves), MFMA / readlane / LDS-DMA counts.
Usage: python tools/isa_table.py file.s"""
import re, sys, collections
fn = None
rows = collections.OrderedDict()
cur = None
for line in open(sys.argv[1]):
    m = re.match(r"^([A-Za-z_][\w.$]*):\s*(;.*)?$", line)
    if m and not m.group(1).startswith(".L"):
        cur = m.group(1)
        rows[cur] = collections.Counter()
        continue
    if cur is None:
        continue
    s = line.strip()
    r = rows[cur]
    if s.startswith("scratch_store"):
        r["st"] += 1
        if "Folded Spill" in s: r["st_spill"] += 1
    elif s.startswith("scratch_load"):
        r["ld"] += 1
        if "Folded Reload" in s: r["ld_spill"] += 1
    elif s.startswith("v_mfma"): r["mfma"] += 1
    elif s.startswith("v_readlane") or s.startswith("v_readfirstlane"): r["readlane"] += 1
    elif s.startswith("global_load_lds"): r["ldsdma"] += 1
    elif s.startswith("global_load") or s.startswith("flat_load"): r["gld"] += 1
    elif s.startswith("global_store") or s.startswith("flat_store"): r["gst"] += 1
    elif s.startswith("ds_"): r["ds"] += 1
    elif s.startswith("s_waitcnt"): r["wait"] += 1
    elif s.startswith("v_"): r["valu"] += 1
    for key, pat in (("len", r"; codeLenInByte = (\d+)"), ("vgpr", r"; NumVgprs: (\d+)"), ("scratch", r"; ScratchSize: (\d+)"),
                     ("sgpr", r"; NumSgprs: (\d+)"), ("occ", r"; Occupancy: (\d+)")):
        mm = re.match(pat, s)
        if mm: r[key] = int(mm.group(1))
print(f"{'function':60s} {'bytes':>7s} {'vgpr':>5s} {'scr B':>6s} {'sc_st':>6s} {'sc_ld':>6s} {'mfma':>5s} {'rdlane':>6s} {'gld':>5s} {'gst':>5s} {'dma':>4s} {'ds':>5s} {'valu':>6s}")
for k, r in rows.items():
    if "len" not in r: continue
    name = k
    try:
        import subprocess
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        pass
    print(f"{name[:60]:60s} {r['len']:7d} {r['vgpr']:5d} {r['scratch']:6d} {r['st']:6d} {r['ld']:6d} {r['mfma']:5d} {r['readlane']:6d} {r['gld']:5d} {r['gst']:5d} {r['ldsdma']:4d} {r['ds']:5d} {r['valu']:6d}")
