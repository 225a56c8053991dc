"""Minimal BIFF8 ``.xls`` writer with the slice of the ``xlwt`` API the reference's
driver uses (``infer.py:75-99``): ``Workbook()``, ``add_sheet(name)``,
``sheet.write(row, col, value)``, ``Workbook.save(path)``.

``xlwt`` is not installed on the MI355X hosts, so the OLE2 compound document and
the BIFF8 record stream are produced here from the published formats ([MS-CFB],
[MS-XLS]).  Strings go through the shared string table (SST / LABELSST) like xlwt
writes them; numbers are written as NUMBER records.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple, Union

Cell = Union[str, int, float]


def _rec(rid: int, data: bytes = b"") -> bytes:
    return struct.pack("<HH", rid, len(data)) + data


def _ustr16(s: str) -> bytes:
    """XLUnicodeString with a 16-bit length (always stored uncompressed UTF-16LE)."""
    return struct.pack("<HB", len(s), 1) + s.encode("utf-16-le")


def _ustr8(s: str) -> bytes:
    """ShortXLUnicodeString (8-bit length)."""
    return struct.pack("<BB", len(s), 1) + s.encode("utf-16-le")


class Worksheet:
    def __init__(self, name: str, book: "Workbook"):
        if not name or len(name) > 31 or any(ch in name for ch in "[]:*?/\\"):
            raise ValueError("invalid worksheet name %r" % name)
        self.name = name
        self._book = book
        self._cells: Dict[Tuple[int, int], Cell] = {}

    def write(self, r: int, c: int, label: Cell = "") -> None:
        if not (0 <= r < 65536 and 0 <= c < 256):
            raise ValueError("cell (%d, %d) out of range for BIFF8" % (r, c))
        if (r, c) in self._cells:
            raise Exception("Attempt to overwrite cell: sheetname=%r rowx=%d colx=%d" % (self.name, r, c))
        if isinstance(label, bool):
            label = int(label)
        if not isinstance(label, (str, int, float)):
            label = str(label)
        self._cells[(r, c)] = label

    def _stream(self) -> bytes:
        out = [_rec(0x0809, struct.pack("<HHHHII", 0x0600, 0x0010, 0x0DBB, 0x07CC, 0, 6))]   # BOF worksheet
        if self._cells:
            rows = [r for r, _ in self._cells]
            cols = [c for _, c in self._cells]
            r0, r1, c0, c1 = min(rows), max(rows) + 1, min(cols), max(cols) + 1
        else:
            r0 = r1 = c0 = c1 = 0
        out.append(_rec(0x0200, struct.pack("<IIHHH", r0, r1, c0, c1, 0)))                    # DIMENSIONS
        by_row: Dict[int, List[int]] = {}
        for (r, c) in self._cells:
            by_row.setdefault(r, []).append(c)
        for r in sorted(by_row):
            cs = by_row[r]
            out.append(_rec(0x0208, struct.pack("<HHHHHHI", r, min(cs), max(cs) + 1, 0x00FF, 0, 0, 0x0F0100)))  # ROW
            for c in sorted(cs):
                v = self._cells[(r, c)]
                if isinstance(v, str):
                    out.append(_rec(0x00FD, struct.pack("<HHHI", r, c, 0x000F, self._book._sst_index(v))))  # LABELSST
                else:
                    out.append(_rec(0x0203, struct.pack("<HHHd", r, c, 0x000F, float(v))))                 # NUMBER
        out.append(_rec(0x023E, struct.pack("<HHHHI", 0x06B6, 0, 0, 0x0040, 0) + b"\x00\x00\x00\x00\x00\x00"))  # WINDOW2
        out.append(_rec(0x000A))                                                                        # EOF
        return b"".join(out)


class Workbook:
    def __init__(self, encoding: str = "ascii"):
        self._sheets: List[Worksheet] = []
        self._sst: Dict[str, int] = {}
        self._sst_list: List[str] = []
        self._sst_refs = 0

    def add_sheet(self, sheetname: str, cell_overwrite_ok: bool = False) -> Worksheet:
        if any(s.name.lower() == sheetname.lower() for s in self._sheets):
            raise Exception("duplicate worksheet name %r" % sheetname)
        ws = Worksheet(sheetname, self)
        self._sheets.append(ws)
        return ws

    def _sst_index(self, s: str) -> int:
        self._sst_refs += 1
        if s not in self._sst:
            self._sst[s] = len(self._sst_list)
            self._sst_list.append(s)
        return self._sst[s]

    # ---- BIFF8 workbook globals
    @staticmethod
    def _font() -> bytes:
        return _rec(0x0031, struct.pack("<HHHHHBBBB", 200, 0, 0x7FFF, 400, 0, 0, 0, 0, 0) + _ustr8("Arial"))

    @staticmethod
    def _xf(style: bool) -> bytes:
        parent = 0xFFF5 if style else 0x0001
        return _rec(0x00E0, struct.pack("<HHHBBBBIIH", 0, 0, parent, 0x20, 0, 0, 0 if style else 0, 0, 0, 0x20C0))

    def _sst_records(self) -> bytes:
        body = struct.pack("<II", self._sst_refs, len(self._sst_list))
        recs = []
        limit = 8224
        for s in self._sst_list:
            enc = _ustr16(s)
            if len(body) + len(enc) > limit:
                # keep every string whole inside one record: start a CONTINUE record
                recs.append(body)
                body = b""
                if len(enc) > limit:
                    raise ValueError("string too long for this minimal SST writer")
            body += enc
        recs.append(body)
        out = _rec(0x00FC, recs[0])
        for extra in recs[1:]:
            out += _rec(0x003C, extra)
        return out

    def _workbook_stream(self) -> bytes:
        if not self._sheets:
            raise IndexError("list index out of range")   # xlwt's behaviour on an empty workbook
        sheet_streams = [s._stream() for s in self._sheets]
        pre = [_rec(0x0809, struct.pack("<HHHHII", 0x0600, 0x0005, 0x0DBB, 0x07CC, 0, 6)),   # BOF globals
               _rec(0x0042, struct.pack("<H", 0x04B0)),                                        # CODEPAGE utf-16
               _rec(0x003D, struct.pack("<HHHHHHHHH", 0x01E0, 0x005A, 0x3FCF, 0x2A4E, 0x0038, 0, 0, 1, 0x0258)),  # WINDOW1
               _rec(0x0022, struct.pack("<H", 0))]                                             # DATEMODE 1900
        pre += [self._font() for _ in range(5)]
        pre += [self._xf(True) for _ in range(15)] + [self._xf(False)]
        pre.append(_rec(0x0293, struct.pack("<HBB", 0x8000, 0, 0xFF)))                         # STYLE Normal
        post = self._sst_records() + _rec(0x000A)
        pre_b = b"".join(pre)
        bs_len = sum(4 + 6 + len(_ustr8(s.name)) for s in self._sheets)
        offset = len(pre_b) + bs_len + len(post)
        bounds = []
        for s, st in zip(self._sheets, sheet_streams):
            bounds.append(_rec(0x0085, struct.pack("<IBB", offset, 0, 0) + _ustr8(s.name)))    # BOUNDSHEET
            offset += len(st)
        return pre_b + b"".join(bounds) + post + b"".join(sheet_streams)

    # ---- OLE2 compound document with one "Workbook" stream
    def save(self, filename) -> None:
        stream = self._workbook_stream()
        # streams shorter than 4096 bytes would have to live in the mini stream: pad instead
        size = max(4096, (len(stream) + 511) // 512 * 512)
        stream = stream.ljust(size, b"\x00")
        n_data = size // 512
        n_fat = 1
        while n_fat * 128 < n_data + 1 + n_fat:
            n_fat += 1
        if n_fat > 109:
            raise ValueError("workbook too large for this minimal writer")
        dir_sect = n_data
        fat_start = n_data + 1
        fat = list(range(1, n_data)) + [0xFFFFFFFE]            # data chain
        fat.append(0xFFFFFFFE)                                   # directory sector
        fat += [0xFFFFFFFD] * n_fat                              # FAT sectors
        fat += [0xFFFFFFFF] * (n_fat * 128 - len(fat))
        header = b"\xD0\xCF\x11\xE0\xA1\xB1\x1A\xE1" + b"\x00" * 16
        header += struct.pack("<HHHHH", 0x003E, 0x0003, 0xFFFE, 9, 6) + b"\x00" * 6
        header += struct.pack("<IIIIIIII", 0, n_fat, dir_sect, 0, 0x1000, 0xFFFFFFFE, 0, 0xFFFFFFFE)
        header += struct.pack("<I", 0)
        difat = [fat_start + i for i in range(n_fat)] + [0xFFFFFFFF] * (109 - n_fat)
        header += struct.pack("<109I", *difat)

        def dirent(name: str, typ: int, child: int, start: int, sz: int) -> bytes:
            raw = name.encode("utf-16-le") + b"\x00\x00"
            e = raw.ljust(64, b"\x00") + struct.pack("<H", len(raw))
            e += struct.pack("<BBIII", typ, 1, 0xFFFFFFFF, 0xFFFFFFFF, child)
            e += b"\x00" * 16 + struct.pack("<I", 0) + b"\x00" * 16
            e += struct.pack("<IQ", start, sz)
            return e

        directory = dirent("Root Entry", 5, 1, 0xFFFFFFFE, 0) + dirent("Workbook", 2, 0xFFFFFFFF, 0, size)
        directory += (b"\x00" * 64 + struct.pack("<H", 0) + struct.pack("<BBIII", 0, 0, 0xFFFFFFFF, 0xFFFFFFFF,
                                                                       0xFFFFFFFF) + b"\x00" * 36 +
                      struct.pack("<IQ", 0, 0)) * 2
        blob = header + stream + directory + struct.pack("<%dI" % len(fat), *fat)
        if hasattr(filename, "write"):
            filename.write(blob)
        else:
            with open(filename, "wb") as f:
                f.write(blob)
