"""Test helper: independent minimal reader for the OLE2 / BIFF8 files roomnet_amd.xls writes
(follows [MS-CFB] / [MS-XLS]; only what the tests need: sheet names, string and number cells)."""
import struct


def workbook_stream(path):
    """The raw BIFF8 `Workbook` stream of an OLE2 compound document."""
    return _open(path)


def read_xls(path):
    stream = _open(path)
    return _decode(stream)


def _open(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\xD0\xCF\x11\xE0\xA1\xB1\x1A\xE1", "not an OLE2 compound document"
    sect_shift, = struct.unpack_from("<H", data, 30)
    assert sect_shift == 9
    n_fat, dir_start = struct.unpack_from("<II", data, 44)
    cutoff, = struct.unpack_from("<I", data, 56)
    difat = struct.unpack_from("<109I", data, 76)

    def sector(i):
        return data[512 + i * 512: 1024 + i * 512]

    fat = []
    for i in range(n_fat):
        fat += list(struct.unpack("<128I", sector(difat[i])))

    def chain(start):
        out, s = b"", start
        while s != 0xFFFFFFFE:
            out += sector(s)
            s = fat[s]
        return out

    directory = chain(dir_start)
    stream = None
    for off in range(0, len(directory), 128):
        ent = directory[off:off + 128]
        nlen, = struct.unpack_from("<H", ent, 64)
        name = ent[:max(nlen - 2, 0)].decode("utf-16-le")
        typ = ent[66]
        start, size = struct.unpack_from("<IQ", ent, 116)
        if typ == 2 and name == "Workbook":
            assert size >= cutoff, "Workbook stream would live in the mini stream"
            stream = chain(start)[:size]
    assert stream is not None, "no Workbook stream"
    return stream


def _decode(stream):

    def records(buf, pos=0):
        while pos + 4 <= len(buf):
            rid, ln = struct.unpack_from("<HH", buf, pos)
            yield pos, rid, buf[pos + 4:pos + 4 + ln]
            pos += 4 + ln
            if rid == 0 and ln == 0:
                return

    def ustr(buf, pos, lenbytes):
        n = struct.unpack_from("<H" if lenbytes == 2 else "<B", buf, pos)[0]
        flags = buf[pos + lenbytes]
        pos += lenbytes + 1
        if flags & 1:
            return buf[pos:pos + 2 * n].decode("utf-16-le"), pos + 2 * n
        return buf[pos:pos + n].decode("latin-1"), pos + n

    sst, sheets = [], []
    sst_buf = None
    for pos, rid, body in records(stream):
        if rid == 0x0085:
            off, = struct.unpack_from("<I", body, 0)
            name, _ = ustr(body, 6, 1)
            sheets.append((name, off))
        elif rid == 0x00FC:
            sst_buf = body
        elif rid == 0x003C and sst_buf is not None:
            sst_buf += body
        elif rid == 0x000A:
            break
    if sst_buf is not None:
        total, uniq = struct.unpack_from("<II", sst_buf, 0)
        p = 8
        for _ in range(uniq):
            s, p = ustr(sst_buf, p, 2)
            sst.append(s)
    out = {}
    for name, off in sheets:
        cells = {}
        first = True
        for pos, rid, body in records(stream, off):
            if first:
                assert rid == 0x0809, "sheet offset does not point at a BOF record"
                first = False
            if rid == 0x00FD:
                r, c, _xf, idx = struct.unpack("<HHHI", body)
                cells[(r, c)] = sst[idx]
            elif rid == 0x0203:
                r, c, _xf, v = struct.unpack("<HHHd", body)
                cells[(r, c)] = v
            elif rid == 0x000A:
                break
        out[name] = cells
    return out
