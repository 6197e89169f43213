"""Build libkgwas_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per ``.hip`` source, compiled in parallel, then one link.  What decides whether an object (and the library) is
current is a CONTENT hash of the source, of every header it can include and of the compile command -- not file times: a fresh
checkout, a copy to another box or a touched file neither trigger nor hide a rebuild.  ``build(force=True)`` (what
``__graft_entry__.build()`` calls: the driver's "does it build" check really compiles) ignores the hashes."""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(CSRC, 'libkgwas_hip.so')
STAMP_DIR = os.path.join(CSRC, 'build')
STAMP = os.path.join(STAMP_DIR, 'stamp.json')          # (what the shipped library was built from: travels with it)
# the objects are a cache, not a product: outside the tree, so that they neither ship to a GPU box with the snapshot nor sit in it
OBJ = os.path.join(tempfile.gettempdir(), 'kgwas_amd_build_' + hashlib.sha256(ROOT.encode()).hexdigest()[:12])
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC']


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join(ROOT, 'include', 'kgwas_hip.h')]


def _digest(paths, extra=''):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _hashes():
    """{source name: hash of (flags, headers, source)} -- every source may include every header of csrc/ and the public one."""
    hd = _digest(_headers(), ' '.join(FLAGS))
    return {os.path.basename(s): _digest([s], hd) for s in sources()}


def _stamp():
    try:
        with open(STAMP) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


# host-side helpers (no GPU code: g++): the result-table writer of KGWAS.train()'s last step, csrc/host/*.cpp
HOST_OUT = os.path.join(CSRC, 'libkgwas_host.so')
HOST_DIR = os.path.join(CSRC, 'host')


def _host_sources():
    return sorted(os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith('.cpp')) if os.path.isdir(HOST_DIR) else []


def build_host(force: bool = False, verbose: bool = True) -> str:
    srcs = _host_sources()
    if not srcs:
        return HOST_OUT
    want = _digest(srcs, 'g++ -O2 -std=c++17')
    st = _stamp()
    if not force and os.path.exists(HOST_OUT) and st.get('host') == want:
        return HOST_OUT
    cmd = [os.environ.get('CXX', 'g++'), '-O2', '-std=c++17', '-fPIC', '-shared', *srcs, '-o', HOST_OUT]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.makedirs(STAMP_DIR, exist_ok=True)
    st['host'] = want
    with open(STAMP, 'w') as f:
        json.dump(st, f)
    return HOST_OUT


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    st = _stamp()
    return st.get('objects') != _hashes() or st.get('lib_size') != os.path.getsize(OUT)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        build_host(False, verbose)
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(STAMP_DIR, exist_ok=True)
    want, have = _hashes(), ({} if force else _stamp().get('objects', {}))
    inc = ['-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
    jobs = []
    for s in sources():
        name = os.path.basename(s)
        o = os.path.join(OBJ, name[:-4] + '.o')
        if force or have.get(name) != want[name] or not os.path.exists(o):
            jobs.append([hipcc, *FLAGS, *inc, '-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + '.o') for s in sources()]
    run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', OUT])
    st = _stamp()
    st.update({'objects': want, 'lib_size': os.path.getsize(OUT)})
    with open(STAMP, 'w') as f:
        json.dump(st, f)
    build_host(force, verbose)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
