"""Minimal GVRS record walker (test-only) used to pull tile packings out of the
reference's binary sample files (tests/golden/ref_samples/*.gvrs).

Layout followed (reference, core/src/main/java/org/gridfour/gvrs/):
  file  : 12-byte id "gvrs raster\\0", version, sub-version, 2 pad bytes, then the
          header record; int32 LE at offset 16 is the header record's size.
          (RecordManager.java:70-78, GvrsFile.java header writer)
  record: int32 LE size (multiple of 8), 1 byte record type (2 = Tile,
          RecordType.java), 3 pad bytes, content (RecordManager.java:161-175)
  tile  : int32 tileIndex, then per element: int32 LE length + bytes
          (RecordManager.java:456-459, RasterTile.java:243-253)
"""
import struct

RECORD_TYPE_TILE = 2


def walk_records(data):
    """Yields (offset, size, type, content_bytes) for every record after the header."""
    assert data[:11] == b"gvrs raster", data[:12]
    (hdr_size,) = struct.unpack_from("<i", data, 16)
    pos = 16 + hdr_size
    while pos + 8 <= len(data):
        (size,) = struct.unpack_from("<i", data, pos)
        rtype = data[pos + 4]
        if size <= 0 or pos + size > len(data):
            break
        yield pos, size, rtype, data[pos + 8:pos + size]
        pos += size


def tile_packings(path, n_elements=1):
    """Returns {tileIndex: [bytes per element]} for every tile record of a file."""
    with open(path, "rb") as f:
        data = f.read()
    out = {}
    for _, _, rtype, content in walk_records(data):
        if rtype != RECORD_TYPE_TILE:
            continue
        (tile_index,) = struct.unpack_from("<i", content, 0)
        p = 4
        elems = []
        for _ in range(n_elements):
            (n,) = struct.unpack_from("<i", content, p)
            p += 4
            elems.append(bytes(content[p:p + n]))
            p += n
        out[tile_index] = elems
    return out
