"""Minimal .xlsx writer / reader (the image has no xlsxwriter / openpyxl).

Enough for the one-sheet error table the entry point writes
(reference: atvsnet/example.py:199-213): inline strings and numbers, one worksheet.
"""
import re
import zipfile
from xml.sax.saxutils import escape

_CT = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>'
       '<Types xmlns="http://schemas.openxmlformats.org/package/2006/content-types">'
       '<Default Extension="rels" ContentType="application/vnd.openxmlformats-package.relationships+xml"/>'
       '<Default Extension="xml" ContentType="application/xml"/>'
       '<Override PartName="/xl/workbook.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.sheet.main+xml"/>'
       '<Override PartName="/xl/worksheets/sheet1.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.worksheet+xml"/>'
       '</Types>')
_RELS = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>'
         '<Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
         '<Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/officeDocument" Target="xl/workbook.xml"/>'
         '</Relationships>')
_WB_RELS = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>'
            '<Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
            '<Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/worksheet" Target="worksheets/sheet1.xml"/>'
            '</Relationships>')


def _col(c):
    s = ''
    c += 1
    while c:
        c, r = divmod(c - 1, 26)
        s = chr(65 + r) + s
    return s


class Workbook(object):
    """workbook = Workbook(path); ws = workbook.add_worksheet(name); ws.write(row, col, value); workbook.close()"""

    def __init__(self, path):
        self.path = path
        self.sheet = None

    def add_worksheet(self, name='Sheet1'):
        self.sheet = _Sheet(name)
        return self.sheet

    def close(self):
        sh = self.sheet or _Sheet('Sheet1')
        rows = {}
        for (r, c), v in sh.cells.items():
            rows.setdefault(r, {})[c] = v
        body = []
        for r in sorted(rows):
            cells = []
            for c in sorted(rows[r]):
                v = rows[r][c]
                ref = '%s%d' % (_col(c), r + 1)
                if isinstance(v, str):
                    cells.append('<c r="%s" t="inlineStr"><is><t>%s</t></is></c>' % (ref, escape(v)))
                else:
                    cells.append('<c r="%s"><v>%s</v></c>' % (ref, repr(float(v))))
            body.append('<row r="%d">%s</row>' % (r + 1, ''.join(cells)))
        sheet_xml = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>'
                     '<worksheet xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main"><sheetData>%s'
                     '</sheetData></worksheet>' % ''.join(body))
        wb_xml = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>'
                  '<workbook xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main" '
                  'xmlns:r="http://schemas.openxmlformats.org/officeDocument/2006/relationships"><sheets>'
                  '<sheet name="%s" sheetId="1" r:id="rId1"/></sheets></workbook>' % escape(sh.name))
        with zipfile.ZipFile(self.path, 'w', zipfile.ZIP_DEFLATED) as z:
            z.writestr('[Content_Types].xml', _CT)
            z.writestr('_rels/.rels', _RELS)
            z.writestr('xl/workbook.xml', wb_xml)
            z.writestr('xl/_rels/workbook.xml.rels', _WB_RELS)
            z.writestr('xl/worksheets/sheet1.xml', sheet_xml)


class _Sheet(object):
    def __init__(self, name):
        self.name = name
        self.cells = {}

    def write(self, row, col, value):
        self.cells[(int(row), int(col))] = value


def read_xlsx(path):
    """-> (sheet_name, {(row, col): str | float}) of the first worksheet (xlsxwriter- or self-written)."""
    with zipfile.ZipFile(path) as z:
        wb = z.read('xl/workbook.xml').decode('utf-8')
        name = re.search(r'<sheet [^>]*name="([^"]*)"', wb).group(1)
        shared = []
        if 'xl/sharedStrings.xml' in z.namelist():
            ss = z.read('xl/sharedStrings.xml').decode('utf-8')
            shared = [re.sub(r'<[^>]+>', '', m) for m in re.findall(r'<si>(.*?)</si>', ss, flags=re.S)]
        sheet = z.read('xl/worksheets/sheet1.xml').decode('utf-8')
    cells = {}
    for attrs, inner in re.findall(r'<c ([^>]*?)(?:/>|>(.*?)</c>)', sheet, flags=re.S):
        ref = re.search(r'r="([A-Z]+)(\d+)"', attrs)
        if not ref or inner is None:
            continue
        col = 0
        for ch in ref.group(1):
            col = col * 26 + (ord(ch) - 64)
        key = (int(ref.group(2)) - 1, col - 1)
        t = re.search(r't="(\w+)"', attrs)
        v = re.search(r'<v>(.*?)</v>', inner, flags=re.S)
        if t and t.group(1) == 's':
            cells[key] = shared[int(v.group(1))]
        elif t and t.group(1) == 'inlineStr':
            cells[key] = re.sub(r'<[^>]+>', '', inner)
        elif v:
            cells[key] = float(v.group(1))
    return name, cells
