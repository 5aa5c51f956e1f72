"""The C++ host above the C ABI (tools/w3cli.cpp, the mirror of src/main.rs:24-87) on the GPU: `w3 <c|d|t> <path>`,
output names, both containers, directory traversal, model selection.  Its streams must be the library's streams."""
import os
import subprocess

import numpy as np
import pytest

import weath3rb0i_amd as w3
from tests.synth import markov_text, mixed_bytes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tools", "w3")


@pytest.fixture(scope="module")
def cli():
    src = os.path.join(ROOT, "tools", "w3cli.cpp")
    if not os.path.exists(CLI) or os.path.getmtime(CLI) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", CLI, src, "-L" + os.path.join(ROOT, "weath3rb0i_amd"), "-lw3hip",
                               "-Wl,-rpath,$ORIGIN/../weath3rb0i_amd", "-Wl,-rpath,/opt/rocm/lib"])
    return CLI


def run(cli, cwd, *args, **env):
    e = dict(os.environ)
    e.update(env)
    return subprocess.run([cli, *args], cwd=cwd, env=e, capture_output=True, text=True, timeout=300)


def parse_block_container(blob):
    assert blob[:4] == b"w3bk" and blob[4] == 1
    orig = int.from_bytes(blob[5:13], "big")
    bs = int.from_bytes(blob[13:17], "big")
    nb = int.from_bytes(blob[17:21], "big")
    lens = [int.from_bytes(blob[21 + 4 * b:25 + 4 * b], "big") for b in range(nb)]
    return orig, bs, lens, blob[21 + 4 * nb:]


@pytest.mark.parametrize("model", ["default", "order012apm", "fullcm"])
def test_cli_test_action_block_container(cli, tmp_path, model):
    data = markov_text(200000, seed=31) + mixed_bytes(70000, seed=32)
    (tmp_path / "in").mkdir()
    f = tmp_path / "in" / "corpus.txt"
    f.write_bytes(data)
    r = run(cli, tmp_path, "t", str(f), W3_MODEL=model)
    assert r.returncode == 0, r.stderr
    assert "Compression took" in r.stdout and "Decompression took" in r.stdout          # main.rs:70-78
    blob = (tmp_path / "corpus.bin").read_bytes()                                        # <name>.bin in the cwd, main.rs:58-68
    assert (tmp_path / "corpus.orig").read_bytes() == data
    orig, bs, lens, body = parse_block_container(blob)
    assert orig == len(data) and bs == 65536 and sum(lens) == len(body)
    ctx = w3.Context(0)
    try:
        m = {"default": w3.init_model, "fullcm": w3.full_cm,
             "order012apm": lambda: w3.APM(w3.BestOfTwoModel(w3.BestOfTwoModel(w3.Order0(), w3.Order1()), w3.OrderN(27, 3)))}[model]()
        out, wl = ctx.encode_blocks(m, data, 65536)
    finally:
        ctx.close()
    assert lens == wl.tolist() and body == out.tobytes()


def test_cli_sharded_compress_gives_the_same_container(cli, tmp_path):
    """W3_SHARDS=3: the C++ host cuts the blocks over three contexts (w3_encode_blocks_sharded, one process): same file bytes."""
    data = markov_text(9 * 65536 + 1234, seed=41)
    f = tmp_path / "corpus.txt"
    f.write_bytes(data)
    assert run(cli, tmp_path, "c", str(f), W3_MODEL="order012apm").returncode == 0
    one = (tmp_path / "corpus.bin").read_bytes()
    r = run(cli, tmp_path, "t", str(f), W3_MODEL="order012apm", W3_SHARDS="3")
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "corpus.bin").read_bytes() == one and (tmp_path / "corpus.orig").read_bytes() == data


def test_cli_reference_container_and_directory(cli, tmp_path, oracle):
    d = tmp_path / "dir"
    d.mkdir()
    a, b = markov_text(30000, seed=5), bytes(range(256)) * 20
    (d / "a.txt").write_bytes(a)
    (d / "b.dat").write_bytes(b)
    (d / "sub").mkdir()                                                                  # shallow traversal: not entered, main.rs:41-50
    r = run(cli, tmp_path, "c", str(d), W3_CONTAINER="w30i")
    assert r.returncode == 0, r.stderr
    blob = (tmp_path / "a.bin").read_bytes()
    assert blob[:4] == b"w30i" and int.from_bytes(blob[4:12], "big") == len(a)           # main.rs:14-15,95-96
    ref = oracle.OrderNEntropy(11, 3, oracle.ACHistory(8, oracle.StationaryModel.for_book1()))
    assert blob == oracle.compress(ref, a)
    r = run(cli, tmp_path, "d", str(tmp_path / "b.bin"))
    assert r.returncode == 0 and (tmp_path / "b.orig").read_bytes() == b


def test_cli_directory_with_files_in_flight(cli, tmp_path):
    """`w3 c <dir>` keeps FILES in flight (w3_encode_host_submit / w3_encode_host_wait: file k+1's input travels and its encode is
    enqueued while file k is coded): every <name>.bin equals what the one-file-at-a-time path (W3_SERIAL=1) writes, empty and tiny
    files included, and decompresses to the file."""
    d = tmp_path / "dir"
    d.mkdir()
    files = {"a.txt": markov_text(300000, seed=51), "b.dat": mixed_bytes(150000, seed=52), "c.txt": markov_text(65536 * 3, seed=53),
             "d.bin": b"", "e.one": b"x", "f.txt": markov_text(70001, seed=54), "g.txt": markov_text(20, seed=55), "h.txt": markov_text(500000, seed=56)}
    for name, data in files.items():
        (d / name).write_bytes(data)
    flight, serial = tmp_path / "flight", tmp_path / "serial"
    flight.mkdir(); serial.mkdir()
    r = run(cli, flight, "c", str(d), W3_MODEL="order012apm")
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("Compression took") == len(files)
    r = run(cli, serial, "c", str(d), W3_MODEL="order012apm", W3_SERIAL="1")
    assert r.returncode == 0, r.stderr
    for name, data in files.items():
        stem = name.split(".")[0]
        a, b = (flight / (stem + ".bin")).read_bytes(), (serial / (stem + ".bin")).read_bytes()
        assert a == b, name
        r = run(cli, flight, "d", str(flight / (stem + ".bin")), W3_MODEL="order012apm")
        assert r.returncode == 0 and (flight / (stem + ".orig")).read_bytes() == data, name


def test_cli_errors(cli, tmp_path):
    assert run(cli, tmp_path, "x", "nothing").returncode == 1                             # usage, main.rs:154-161
    assert run(cli, tmp_path, "c", str(tmp_path / "missing")).returncode == 1
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"not a container at all....")
    r = run(cli, tmp_path, "d", str(bad))
    assert r.returncode == 1 and "Magic numbers" in r.stderr


def test_cli_large_file_goes_through_in_pieces(cli, tmp_path):
    """One library call handles less than 4 GiB; the CLI cuts a larger file into calls of whole blocks (W3_CALL_MAX: the piece size,
    2 GiB by default).  Blocks are independent, so the container must be the same bytes whatever the pieces — and decode back,
    also piece by piece."""
    data = markov_text(300000, seed=41) + mixed_bytes(123457, seed=42)
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    for d in ("a", "b"):
        (tmp_path / d / "big.txt").write_bytes(data)
    r = run(cli, tmp_path / "a", "c", "big.txt", W3_MODEL="order012apm")
    assert r.returncode == 0, r.stderr
    r = run(cli, tmp_path / "b", "t", "big.txt", W3_MODEL="order012apm", W3_CALL_MAX="131072")   # seven calls (two blocks each, a ragged last one)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "a" / "big.bin").read_bytes() == (tmp_path / "b" / "big.bin").read_bytes()
    assert (tmp_path / "b" / "big.orig").read_bytes() == data
    r = run(cli, tmp_path / "a", "d", "big.bin", W3_MODEL="order012apm", W3_CALL_MAX="65536")
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "a" / "big.orig").read_bytes() == data
