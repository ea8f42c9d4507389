"""The extension's registering allocator as an ARENA (infera_extension_hip.cpp InferaAllocatorData, round 6): blocks of the buffer manager's
size come out of 64 MiB slabs of 256 blocks, lowest free block of the oldest slab with room first; other sizes keep the default path.  Driven
call by call through the DuckDB stand-in -- no GPU needed: a registration that fails only leaves the memory unregistered."""
import ctypes as C
import threading

import pytest

BLOCK = 262144
SLAB = 256 * BLOCK


@pytest.fixture()
def alloc(built, monkeypatch):
    from infera_amd import sqlharness

    monkeypatch.setenv("INFERA_ZERO_COPY_ALLOCATOR", "1")
    monkeypatch.delenv("INFERA_ZERO_COPY_ARENA", raising=False)
    L = sqlharness.lib()
    L.infera_stub_allocator_create.argtypes = [C.POINTER(C.c_int32)]
    L.infera_stub_allocator_create.restype = C.c_void_p
    L.infera_stub_allocator_destroy.argtypes = [C.c_void_p]
    L.infera_stub_allocate.argtypes = [C.c_void_p, C.c_uint64]
    L.infera_stub_allocate.restype = C.c_void_p
    L.infera_stub_free.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.infera_stub_reallocate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
    L.infera_stub_reallocate.restype = C.c_void_p
    hooked = C.c_int32()
    h = L.infera_stub_allocator_create(C.byref(hooked))
    assert hooked.value == 1
    yield L, h
    L.infera_stub_allocator_destroy(h)


def test_blocks_come_out_of_slabs_in_address_order_and_freed_ones_are_reused_lowest_first(alloc):
    L, h = alloc
    p = [L.infera_stub_allocate(h, BLOCK) for _ in range(600)]
    assert all(p) and len(set(p)) == 600
    # the first 256 at one stride inside one 2 MiB-aligned slab; then the next slab
    assert p[0] % (2 << 20) == 0 and [q - p[0] for q in p[:256]] == [i * BLOCK for i in range(256)]
    assert p[256] % (2 << 20) == 0 and [q - p[256] for q in p[256:512]] == [i * BLOCK for i in range(256)]
    assert not (p[0] <= p[256] < p[0] + SLAB)
    # blocks are writable end to end
    C.memset(p[599], 0xAB, BLOCK)
    # free a few in the first slab: the next allocations take exactly those, lowest first
    for i in (7, 3, 200):
        L.infera_stub_free(h, p[i], BLOCK)
    again = [L.infera_stub_allocate(h, BLOCK) for _ in range(3)]
    assert again == [p[3], p[7], p[200]]
    # empty the third slab (blocks 512..599) and the second: one empty slab is kept, so the next allocation reuses an existing slab's first block
    for q in p[256:]:
        L.infera_stub_free(h, q, BLOCK)
    nxt = L.infera_stub_allocate(h, BLOCK)
    assert nxt in (p[256], p[512])
    for q in p[:256] + [nxt]:
        L.infera_stub_free(h, q, BLOCK)


def test_other_sizes_keep_the_default_path_and_reallocate_moves_between_the_two(alloc):
    L, h = alloc
    small = L.infera_stub_allocate(h, 4096)
    big = L.infera_stub_allocate(h, 3 * BLOCK)        # >= 128 KiB but not the arena's size: its own (attempted) registration
    blk = L.infera_stub_allocate(h, BLOCK)
    assert small and big and blk and blk % (2 << 20) == 0
    C.memset(blk, 0x5A, BLOCK)
    grown = L.infera_stub_reallocate(h, blk, BLOCK, 2 * BLOCK)          # out of the arena
    assert grown and grown != blk and C.string_at(grown, 64) == b"\x5a" * 64
    assert L.infera_stub_allocate(h, BLOCK) == blk                      # ... and its slab block is free again
    shrunk = L.infera_stub_reallocate(h, grown, 2 * BLOCK, BLOCK)       # into the arena
    assert shrunk and shrunk == blk + BLOCK and C.string_at(shrunk, 64) == b"\x5a" * 64
    L.infera_stub_free(h, small, 4096)
    L.infera_stub_free(h, big, 3 * BLOCK)
    L.infera_stub_free(h, shrunk, BLOCK)
    L.infera_stub_free(h, blk, BLOCK)


def test_arena_under_threads(alloc):
    L, h = alloc
    errors = []

    def worker(seed):
        mine = []
        x = seed
        for _ in range(4000):
            x = (x * 6364136223846793005 + 1442695040888963407) & (2**64 - 1)
            if len(mine) < 40 and (x >> 33) % 3 != 0:
                q = L.infera_stub_allocate(h, BLOCK)
                if not q:
                    errors.append("allocation failed")
                    return
                C.memset(q, seed, 64)
                mine.append(q)
            elif mine:
                q = mine.pop((x >> 40) % len(mine))
                if C.string_at(q, 64) != bytes([seed]) * 64:
                    errors.append("a block was handed to two owners")
                L.infera_stub_free(h, q, BLOCK)
        for q in mine:
            L.infera_stub_free(h, q, BLOCK)

    ts = [threading.Thread(target=worker, args=(i + 1,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[:3]


def test_arena_switch_off_restores_plain_blocks(built, monkeypatch):
    from infera_amd import sqlharness

    monkeypatch.setenv("INFERA_ZERO_COPY_ALLOCATOR", "1")
    monkeypatch.setenv("INFERA_ZERO_COPY_ARENA", "0")
    L = sqlharness.lib()
    L.infera_stub_allocator_create.argtypes = [C.POINTER(C.c_int32)]
    L.infera_stub_allocator_create.restype = C.c_void_p
    L.infera_stub_allocate.argtypes = [C.c_void_p, C.c_uint64]
    L.infera_stub_allocate.restype = C.c_void_p
    L.infera_stub_free.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.infera_stub_allocator_destroy.argtypes = [C.c_void_p]
    hooked = C.c_int32()
    h = L.infera_stub_allocator_create(C.byref(hooked))
    try:
        p = [L.infera_stub_allocate(h, BLOCK) for _ in range(4)]
        assert all(p) and [q - p[0] for q in p] != [i * BLOCK for i in range(4)]  # malloc'ed: not a slab's stride
        for q in p:
            L.infera_stub_free(h, q, BLOCK)
    finally:
        L.infera_stub_allocator_destroy(h)
