"""The C-ABI library loads on a machine without a GPU and exports every symbol the header declares."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="gr80211_rx.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gr80211_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(pkg):
    names = declared_functions()
    assert len(names) >= 14
    lib = pkg.load_library()
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(pkg.rx.EXPORTS) == names


def test_struct_sizes_match_header(pkg):
    assert ctypes.sizeof(pkg.rx.FrameDesc) == 32
    assert ctypes.sizeof(pkg.rx.FrameInfo) == 112
    assert ctypes.sizeof(pkg.rx.DecDesc) == 40
    assert pkg.rx.SIG_DESC_DTYPE.itemsize == 48 and pkg.rx.TX_DESC_DTYPE.itemsize == 40 and pkg.rx.TX_MU_DESC_DTYPE.itemsize == 56
    # the C compiler lays gr80211_sig_desc out as the binding does: 8-byte offset first, 44 bytes of fields, padded to 48
    names = [n for n in pkg.rx.SIG_DESC_DTYPE.names]
    assert names[:10] == ["off", "n_avail", "mcs", "len", "nsamp", "cfo", "snr", "rssi", "mu_pos", "reserved"]
    assert pkg.rx.SIG_DESC_DTYPE.fields["mu_pos"][1] == 36


def test_version_and_tables(pkg):
    lib = pkg.load_library()
    assert b"gfx950" in lib.gr80211_version()
    assert lib.gr80211_table_checksum() not in (0, 0xFFFFFFFF)


def test_null_context_is_rejected(pkg):
    lib = pkg.load_library()
    assert lib.gr80211_sync(None, None) == -1
    assert lib.gr80211_set_profiling(None, 1) == -1
    assert lib.gr80211_last_error(None) == b"null context"


def test_host_blocks_library_loads():
    """The C++ host blocks link against the C-ABI library and load without a GPU; the command-line
    scheduler rejects a bad invocation before it touches the device."""
    import subprocess
    d = os.path.join(ROOT, "fork-gr-ieee80211_amd")
    assert os.path.exists(os.path.join(d, "libgr80211_host.so")), "run __graft_entry__.build()"
    ctypes.CDLL(os.path.join(d, "libgr80211_rx.so"), mode=ctypes.RTLD_GLOBAL)
    ctypes.CDLL(os.path.join(d, "libgr80211_host.so"))
    p = subprocess.run([os.path.join(d, "gr80211_pump")], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "usage" in p.stderr


def test_streaming_abi_symbols_exported(pkg):
    """Every function include/gr80211_host.h declares is exported by libgr80211_host.so; an unknown block kind is
    refused with a reason (no device is touched for it)."""
    from importlib import import_module
    stream = import_module("fork-gr-ieee80211_amd.stream")
    names = [n for n in declared_functions("gr80211_host.h") if n.startswith("gr80211_block_")]
    assert sorted(stream.EXPORTS) == names
    lib = stream.load_host_library()
    for n in names:
        assert hasattr(lib, n), n
    for n in stream.UDP_EXPORTS:
        assert n in declared_functions("gr80211_host.h") and hasattr(lib, n), n
    assert ctypes.sizeof(stream.Tag) == 40
    assert not lib.gr80211_block_create(99, 0, 0)
    assert b"unknown block kind" in lib.gr80211_block_last_error(None)
    assert lib.gr80211_block_n_inputs(None) == -1 and lib.gr80211_block_flush(None) == -1


def test_udp_sender_one_datagram_per_blob(pkg):
    """The UDP hand-off of the reference flowgraphs (socket_pdu UDP_CLIENT behind decode's "out" port,
    examples/cmu_v3/trxSta0.grc:301-313,1054): every blob [format][len lo][len hi][payload][mcs] arrives as one
    datagram with exactly its bytes, in order on the loopback; an empty blob and one of the largest size a datagram
    carries pass too, a longer one is cut to that size.  No GPU involved."""
    import socket
    from importlib import import_module
    stream = import_module("fork-gr-ieee80211_amd.stream")
    lib = stream.load_host_library()
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    rx.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 1 << 22)
    rx.bind(("127.0.0.1", 0))
    rx.settimeout(5.0)
    port = rx.getsockname()[1]
    assert not lib.gr80211_udp_open(b"127.0.0.1", 0) and not lib.gr80211_udp_open(None, port)
    assert lib.gr80211_udp_send(None, b"x", 1) == -1
    u = lib.gr80211_udp_open(b"127.0.0.1", port)
    assert u
    rng = np.random.default_rng(5)
    blobs = []
    for fmt, n, mcs in ((0, 100, 3), (1, 1500, 7), (2, 4095, 8), (2, 1, 0)):
        payload = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        blobs.append(bytes([fmt, n & 255, n >> 8]) + payload + bytes([mcs]))
    blobs.append(b"")
    blobs.append(bytes(65507))
    try:
        for b in blobs:
            assert lib.gr80211_udp_send(u, b, len(b)) == 0
            got, _ = rx.recvfrom(1 << 17)
            assert got == b
            if len(b) > 3:
                assert len(got) == 3 + (got[1] | got[2] << 8) + 1 or got[0] == 0 and len(got) == 65507
        long_one = bytes(range(256)) * 300
        assert lib.gr80211_udp_send(u, long_one, len(long_one)) == 0
        got, _ = rx.recvfrom(1 << 17)
        assert got == long_one[:65507]
    finally:
        lib.gr80211_udp_close(u)
        rx.close()


def test_tx_frame_lengths_match_generator(pkg):
    """gr80211_tx_samples is host arithmetic (no GPU): the frame lengths equal the numpy generator's bookkeeping,
    which test_txgen.py pins to the reference generator's waveforms; unsupported combinations give -1."""
    lib = pkg.load_library()
    tx = pkg.txgen
    for fmt, mcss, nsss in ((0, range(8), (1,)), (1, range(16), (1,)), (2, range(9), (1, 2))):
        for mcs in mcss:
            for nss in nsss:
                for ln in (1, 2, 13, 100, 1000, 1500, 4095):
                    m = tx.Mod(fmt, mcs, nss).set_len(ln)
                    want = m.n_samples() if m.lsig_len <= 4095 else -1
                    assert lib.gr80211_tx_samples(fmt, mcs, nss, ln) == want, (fmt, mcs, nss, ln)
    for bad in ((0, 8, 1, 100), (1, 16, 1, 100), (2, 9, 1, 100), (2, 0, 3, 100), (3, 0, 1, 100), (0, 0, 1, 0), (1, 3, 1, 0), (0, 0, 1, 4096)):
        assert lib.gr80211_tx_samples(*bad) == -1, bad
    assert lib.gr80211_tx_samples(2, 0, 2, 0) == 880 and lib.gr80211_tx_samples(2, 0, 1, 0) == 800
    for bad in ():
        assert lib.gr80211_tx_samples(*bad) == -1, bad
    assert ctypes.sizeof(ctypes.c_char * pkg.rx.TX_DESC_DTYPE.itemsize) == 40


def test_large_batch_decoder_leaves_register_room(pkg, tmp_path):
    """Two waves of the one-wave decoder of large batches must leave a SIMD's register file (512 per lane) room for a
    72-register wave of the next batch's preamble or demodulator kernel: at most 216 registers.  The limit is a
    function attribute whose unit the compiler defines (it counts in pairs on this target), so the result is read
    from the code object the library carries."""
    import shutil
    import subprocess
    tools = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(os.path.join(tools, "llvm-objdump")) and os.path.exists(os.path.join(tools, "llvm-readelf"))):
        import pytest
        pytest.skip("no llvm-objdump / llvm-readelf on this machine")
    so = tmp_path / "lib.so"
    shutil.copy(pkg.rx.LIB_PATH, so)
    subprocess.run([os.path.join(tools, "llvm-objdump"), "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    found = {}
    for f in sorted(tmp_path.glob("lib.so.*gfx950*")):
        notes = subprocess.run([os.path.join(tools, "llvm-readelf"), "--notes", str(f)], capture_output=True, text=True).stdout
        for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.vgpr_count:\s+(\d+)", notes, flags=re.S):
            if "\n    .name:" in m.group(2):
                continue  # ran into the next kernel's record
            found[m.group(1)] = int(m.group(3))
    lean = [v for k, v in found.items() if "k_decode_leanE" in k]
    demod = [v for k, v in found.items() if "k_demodILi1ELb0E" in k]
    pre = [v for k, v in found.items() if "k_preamble" in k]
    assert len(lean) == 1 and len(demod) == 1 and len(pre) == 1, sorted(found)
    gran = 8  # registers are handed out in blocks of eight
    up = lambda v: (v + gran - 1) // gran * gran
    assert 2 * up(lean[0]) + up(demod[0]) <= 512 and 2 * up(lean[0]) + up(pre[0]) <= 512, (lean, demod, pre)


def test_lane_state_decoder_leaves_its_look_ahead_registers_alone(pkg, tmp_path):
    """The one-wave-per-frame decoder brings soft bits into L2 with loads issued from inline assembly, which the
    compiler does not count: nothing may write such a load's destination register before the wait that follows it, or
    the late arrival would overwrite a live value.  Read from the code object the library carries: every single-dword
    global load without a scalar base in that kernel, up to the next full wait for vector memory in
    program order (the loads inside the loops are waited for in the same basic block)."""
    import shutil
    import subprocess
    tools = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(tools, "llvm-objdump")):
        import pytest
        pytest.skip("no llvm-objdump on this machine")
    so = tmp_path / "lib.so"
    shutil.copy(pkg.rx.LIB_PATH, so)
    subprocess.run([os.path.join(tools, "llvm-objdump"), "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    text = ""
    for f in sorted(tmp_path.glob("lib.so.*gfx950*")):
        text += subprocess.run([os.path.join(tools, "llvm-objdump"), "-d", str(f)], capture_output=True, text=True).stdout
    # (the instantiation the library launches by itself: <false>, soft bits by scalar loads)
    m = re.search(r"^[0-9a-f]+ <(\S*k_decode_lane_stateILb0E\S*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", text, flags=re.S | re.M)
    assert m, "kernel not found in the code object"
    ins = [l.split("//")[0].strip() for l in m.group(2).split("\n")]
    ins = [l for l in ins if l and not l.endswith(":")]

    def written(line):
        op, _, rest = line.partition(" ")
        if op.startswith(("global_store", "buffer_store", "ds_write", "v_cmp", "s_", "v_readlane", "v_readfirstlane")):
            return set()
        dst = rest.split(",")[0].strip()
        regs = set()
        a = re.fullmatch(r"v(\d+)", dst)
        b = re.fullmatch(r"v\[(\d+):(\d+)\]", dst)
        if a:
            regs = {int(a.group(1))}
        if b:
            regs = set(range(int(b.group(1)), int(b.group(2)) + 1))
        if op.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap")):
            regs |= {int(x) for x in re.findall(r"\bv(\d+)\b", rest)}
        return regs

    loads = [i for i, l in enumerate(ins) if re.match(r"global_load_dword v\d+, v\[\d+:\d+\], off", l)]
    assert len(loads) >= 12, loads      # four code rates: one before the loops, one in each of the two loops
    for i in loads:
        reg = int(re.match(r"global_load_dword v(\d+),", ins[i]).group(1))
        j = i + 1
        while j < len(ins) and not ins[j].startswith("s_waitcnt vmcnt(0)"):
            assert reg not in written(ins[j]), (ins[i], ins[j])
            assert not ins[j].startswith("s_endpgm"), (ins[i], "no wait before the code ends")
            j += 1
        assert j < len(ins)
