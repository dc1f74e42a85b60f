"""Static per-function table of a gfx950 assembly file (hipcc -save-temps ... .s): code size, VGPRs, scratch size,
scratch loads / stores - split into those at the entry / exit of an outlined function (saves and restores of callee-saved
registers: executed once per call of the phase) and those in its body -, MFMA / readlane / LDS-DMA counts.
Usage: python tools/isa_table.py file.s"""
import re, sys, collections
fn = None
rows = collections.OrderedDict()
pos = {}
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
    r["n"] += 1
    if s.startswith("scratch_store"):
        r["st"] += 1
        pos.setdefault(cur, []).append(r["n"])
    elif s.startswith("scratch_load"):
        r["ld"] += 1
        pos.setdefault(cur, []).append(r["n"])
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
print(f"{'function':60s} {'bytes':>7s} {'vgpr':>5s} {'scr B':>6s} {'sc_st':>6s} {'sc_ld':>6s} {'body':>5s} {'mfma':>5s} {'rdlane':>6s} {'gld':>5s} {'gst':>5s} {'dma':>4s} {'ds':>5s} {'valu':>6s}")
for k, r in rows.items():
    if "len" not in r: continue
    name = k
    try:
        import subprocess
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        pass
    # scratch operations away from the entry / exit of an outlined function (a kernel has no callee-saved registers: all body)
    n = r["n"]
    edge = 0 if "kernel" in k and not k.startswith("_Z") else max(260, n // 25)
    body = sum(1 for q in pos.get(k, []) if edge < q < n - edge) if edge else len(pos.get(k, []))
    print(f"{name[:60]:60s} {r['len']:7d} {r['vgpr']:5d} {r['scratch']:6d} {r['st']:6d} {r['ld']:6d} {body:5d} {r['mfma']:5d} {r['readlane']:6d} {r['gld']:5d} {r['gst']:5d} {r['ldsdma']:4d} {r['ds']:5d} {r['valu']:6d}")
