"""tools/check_rot_isa.py -- the build's check of the rotation kernels' generated code -- on doctored listings: a DPP
hazard is caught and turned into the FMMBEM_ROT_NOP_ORDERS mask the Makefile rebuilds with (csrc/Makefile ROTBUILD), the
cured listing passes, and an early touch of a load in flight is reported as something no wait state cures."""
import os
import subprocess
import sys

from conftest import ROOT

CHECK = os.path.join(ROOT, "tools", "check_rot_isa.py")
HEAD = "0000000000001000 <_ZN6fmmbem12_GLOBAL__N_114m2l_rot_kernelILi%dEEEvNS_10DevicePlanENS_7RotWorkE>:\n"


def listing(tmp_path, name, order, body):
    text = HEAD % order + "".join("\t%s // %012X: 00000000\n" % (ins, 0x1000 + 8 * k) for k, ins in enumerate(body))
    f = tmp_path / name
    f.write_text(text)
    return str(f)


def run(*args):
    return subprocess.run([sys.executable, CHECK, *args], capture_output=True, text=True)


DPP = "v_fmac_f64_dpp v[10:11], v[4:5], v[12:13] row_newbcast:3 row_mask:0xf bank_mask:0xf"


def test_dpp_hazard_is_caught_and_named(tmp_path):
    bad = listing(tmp_path, "bad.s", 7, ["v_mov_b64_e32 v[4:5], v[8:9]", DPP, "s_endpgm"])
    r = run("--text", bad)
    assert r.returncode == 1 and "p=7" in r.stdout and "dpp hazards 1" in r.stdout and "early touches of loads in flight 0" in r.stdout
    rep = tmp_path / "bad.isa"
    rep.write_text(r.stdout)
    m = run("--nop-mask", str(rep))
    assert m.returncode == 0 and m.stdout.strip() == "0x40"            # bit p - 1


def test_wait_states_cure_it(tmp_path):
    for body in (["v_mov_b64_e32 v[4:5], v[8:9]", "s_nop 1", DPP, "s_endpgm"],                      # what kNop builds emit
                 ["v_mov_b64_e32 v[4:5], v[8:9]", "v_add_f64 v[20:21], v[22:23], v[24:25]", "v_add_f64 v[26:27], v[22:23], v[24:25]", DPP, "s_endpgm"],
                 ["v_mov_b64_e32 v[6:7], v[8:9]", DPP, "s_endpgm"]):                                   # another register
        r = run("--text", listing(tmp_path, "ok.s", 7, body))
        assert r.returncode == 0 and "dpp hazards 0" in r.stdout, r.stdout


def test_early_touch_of_a_load_in_flight_is_not_curable(tmp_path):
    body = ["global_load_dwordx2 v[20:21], v[2:3], off", "v_add_f64 v[30:31], v[20:21], v[22:23]", "s_waitcnt vmcnt(0)", DPP, "s_endpgm"]
    r = run("--text", listing(tmp_path, "early.s", 9, body))
    assert r.returncode == 1 and "early touches of loads in flight 1" in r.stdout
    rep = tmp_path / "early.isa"
    rep.write_text(r.stdout)
    assert run("--nop-mask", str(rep)).stdout.strip() == "0x0"
    # the same load waited for before its use is fine
    body = ["global_load_dwordx2 v[20:21], v[2:3], off", "s_waitcnt vmcnt(0)", "v_add_f64 v[30:31], v[20:21], v[22:23]", DPP, "s_endpgm"]
    assert run("--text", listing(tmp_path, "late.s", 9, body)).returncode == 0


def test_makefile_rebuilds_once_with_the_mask(tmp_path):
    """csrc/Makefile ROTBUILD with a stub compiler and a stub checker: a DPP-hazard report makes it rebuild the object once with
    -DFMMBEM_ROT_NOP_ORDERS=<mask> and pass; a report with an early touch fails the build and leaves no object behind."""
    import shutil
    csrc = tmp_path / "pkg" / "lib" / "csrc"
    csrc.mkdir(parents=True)
    (tmp_path / "pkg" / "tools").mkdir()
    shutil.copy(os.path.join(ROOT, "fmm-bem-relaxed_amd", "csrc", "Makefile"), csrc / "Makefile")
    shutil.copy(CHECK, tmp_path / "pkg" / "tools" / "check_rot_isa.py")
    for f in ("kernels_m2l_rot.hip", "device_plan.hpp", "device_launch.hpp", "m2l_rot.hpp"):
        (csrc / f).write_text("")
    cc, chk, log, cnt = tmp_path / "cc.sh", tmp_path / "chk.sh", tmp_path / "cc.log", tmp_path / "count"
    cc.write_text('#!/bin/sh\nout=""; prev=""\nfor a in "$@"; do [ "$prev" = "-o" ] && out="$a"; prev="$a"; done\necho "$@" >> %s\necho obj > "$out"\n' % log)
    chk.write_text('#!/bin/sh\nn=$(cat %s 2>/dev/null || echo 0); echo $((n+1)) > %s\n'
                   'if [ "$n" = "0" ]; then echo "p=10  dpp  1741  dpp hazards 3  early touches of loads in flight 0"; exit 1; fi\n'
                   'echo "p=10  dpp  1741  dpp hazards 0  early touches of loads in flight 0"; exit 0\n' % (cnt, cnt))
    os.chmod(cc, 0o755)
    os.chmod(chk, 0o755)
    r = subprocess.run(["make", "kernels_m2m_rot.o", "ROTCC=%s" % cc, "ROTCHECK=%s" % chk], cwd=csrc, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    calls = log.read_text().splitlines()
    assert len(calls) == 2 and "FMMBEM_ROT_NOP_ORDERS" not in calls[0] and "-DFMMBEM_ROT_NOP_ORDERS=0x200u" in calls[1]
    assert all("-DFMMBEM_ROT_OP=1" in c for c in calls)
    assert (csrc / "kernels_m2m_rot.o").exists()
    os.remove(csrc / "kernels_m2m_rot.o")
    chk.write_text('#!/bin/sh\necho "p=9  dpp  1277  dpp hazards 0  early touches of loads in flight 2"; exit 1\n')
    r = subprocess.run(["make", "kernels_m2m_rot.o", "ROTCC=%s" % cc, "ROTCHECK=%s" % chk], cwd=csrc, capture_output=True, text=True)
    assert r.returncode != 0 and "no wait state cures" in r.stdout and not (csrc / "kernels_m2m_rot.o").exists()
