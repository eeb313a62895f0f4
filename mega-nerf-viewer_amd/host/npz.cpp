#include "npz.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace viewer::npz {

namespace {

uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t rd64(const uint8_t *p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

struct File {
    std::FILE *fp;
    explicit File(const std::string &path) : fp(std::fopen(path.c_str(), "rb")) {
        if (!fp) throw std::runtime_error("npz: cannot open " + path);
    }
    ~File() { std::fclose(fp); }
    uint64_t size() {
        fseeko(fp, 0, SEEK_END);
        return (uint64_t)ftello(fp);
    }
    void read_at(uint64_t off, void *dst, size_t n) {
        if (fseeko(fp, (off_t)off, SEEK_SET) != 0 || std::fread(dst, 1, n, fp) != n)
            throw std::runtime_error("npz: truncated file");
    }
};

// Walk a ZIP "extra field" block for the ZIP64 record (id 0x0001) and patch the
// 32-bit fields that were saturated to 0xFFFFFFFF, in the order the spec lists them.
void apply_zip64_extra(const uint8_t *extra, size_t len, uint64_t &usize, uint64_t &csize, uint64_t &lho) {
    size_t i = 0;
    while (i + 4 <= len) {
        const uint16_t id = rd16(extra + i), sz = rd16(extra + i + 2);
        if (i + 4 + (size_t)sz > len) return;  // record runs past the block: ignore it
        if (id == 0x0001) {
            const uint8_t *q = extra + i + 4;
            size_t left = sz;
            if (usize == 0xFFFFFFFFull && left >= 8) { usize = rd64(q); q += 8; left -= 8; }
            if (csize == 0xFFFFFFFFull && left >= 8) { csize = rd64(q); q += 8; left -= 8; }
            if (lho == 0xFFFFFFFFull && left >= 8) { lho = rd64(q); }
            return;
        }
        i += 4 + (size_t)sz;
    }
}

std::string dict_value(const std::string &hdr, const std::string &key) {
    size_t k = hdr.find("'" + key + "'");
    if (k == std::string::npos) throw std::runtime_error("npy: header lacks '" + key + "'");
    k = hdr.find(':', k);
    if (k == std::string::npos) throw std::runtime_error("npy: malformed header");
    size_t b = k + 1;
    while (b < hdr.size() && hdr[b] == ' ') ++b;
    size_t e = b;
    if (hdr[b] == '(') {
        e = hdr.find(')', b);
        if (e == std::string::npos) throw std::runtime_error("npy: malformed shape");
        return hdr.substr(b, e - b + 1);
    }
    if (hdr[b] == '\'') {
        e = hdr.find('\'', b + 1);
        return hdr.substr(b + 1, e - b - 1);
    }
    while (e < hdr.size() && hdr[e] != ',' && hdr[e] != '}') ++e;
    return hdr.substr(b, e - b);
}

}  // namespace

Array parse_npy(const uint8_t *buf, size_t len) {
    if (len < 10 || std::memcmp(buf, "\x93NUMPY", 6) != 0) throw std::runtime_error("npy: bad magic");
    const int major = buf[6];
    size_t hlen, hoff;
    if (major == 1) {
        hlen = rd16(buf + 8);
        hoff = 10;
    } else {
        if (len < 12) throw std::runtime_error("npy: truncated header");
        hlen = rd32(buf + 8);
        hoff = 12;
    }
    if (hoff + hlen > len) throw std::runtime_error("npy: truncated header");
    const std::string hdr(reinterpret_cast<const char *>(buf + hoff), hlen);

    Array a;
    const std::string descr = dict_value(hdr, "descr");
    if (descr.size() < 2) throw std::runtime_error("npy: unsupported descr " + descr);
    size_t p = 0;
    if (descr[0] == '<' || descr[0] == '|' || descr[0] == '=') p = 1;
    else if (descr[0] == '>') throw std::runtime_error("npy: big-endian arrays are not supported");
    a.kind = descr[p];
    const size_t n = (size_t)std::strtoul(descr.c_str() + p + 1, nullptr, 10);
    a.word_size = (a.kind == 'U') ? 4 * n : n;
    a.fortran_order = dict_value(hdr, "fortran_order").rfind("True", 0) == 0;
    const std::string shape = dict_value(hdr, "shape");
    for (size_t i = 1; i < shape.size();) {
        while (i < shape.size() && (shape[i] == ' ' || shape[i] == ',')) ++i;
        if (i >= shape.size() || shape[i] == ')') break;
        char *end = nullptr;
        a.shape.push_back((size_t)std::strtoull(shape.c_str() + i, &end, 10));
        if (end == shape.c_str() + i) throw std::runtime_error("npy: malformed shape " + shape);
        i = (size_t)(end - shape.c_str());
    }
    // element count and byte size with overflow checks: a header must not describe more than the payload holds
    const size_t payload = len - hoff - hlen;
    size_t nbytes = a.word_size;
    if (a.word_size == 0) throw std::runtime_error("npy: unsupported descr " + descr);
    for (size_t dim : a.shape) {
        if (dim != 0 && nbytes > payload / dim) throw std::runtime_error("npy: truncated payload");
        nbytes *= dim;
    }
    if (nbytes > payload) throw std::runtime_error("npy: truncated payload");
    a.bytes.assign(buf + hoff + hlen, buf + hoff + hlen + nbytes);
    return a;
}

Archive load(const std::string &path) {
    File f(path);
    const uint64_t fsize = f.size();
    if (fsize < 22) throw std::runtime_error("npz: file too small");

    // End-of-central-directory record: scan the last 64 KiB + 22 bytes backwards.
    const uint64_t tail = fsize < 65557 ? fsize : 65557;
    std::vector<uint8_t> buf(tail);
    f.read_at(fsize - tail, buf.data(), tail);
    int64_t eocd = -1;
    for (int64_t i = (int64_t)tail - 22; i >= 0; --i) {
        if (rd32(buf.data() + i) == 0x06054b50u) { eocd = i; break; }
    }
    if (eocd < 0) throw std::runtime_error("npz: no end-of-central-directory record");
    uint64_t n_entries = rd16(buf.data() + eocd + 10);
    uint64_t cd_size = rd32(buf.data() + eocd + 12);
    uint64_t cd_off = rd32(buf.data() + eocd + 16);
    if (n_entries == 0xFFFF || cd_size == 0xFFFFFFFFull || cd_off == 0xFFFFFFFFull) {
        // ZIP64: locator (20 bytes) sits right before the EOCD
        if (eocd < 20 || rd32(buf.data() + eocd - 20) != 0x07064b50u) throw std::runtime_error("npz: missing ZIP64 locator");
        const uint64_t z64 = rd64(buf.data() + eocd - 20 + 8);
        uint8_t rec[56];
        f.read_at(z64, rec, sizeof(rec));
        if (rd32(rec) != 0x06064b50u) throw std::runtime_error("npz: bad ZIP64 end record");
        n_entries = rd64(rec + 32);
        cd_size = rd64(rec + 40);
        cd_off = rd64(rec + 48);
    }
    if (cd_off > fsize || cd_size > fsize - cd_off) throw std::runtime_error("npz: central directory lies outside the file");
    std::vector<uint8_t> cd(cd_size);
    f.read_at(cd_off, cd.data(), cd_size);

    Archive out;
    size_t p = 0;
    for (uint64_t e = 0; e < n_entries; ++e) {
        if (p + 46 > cd.size() || rd32(cd.data() + p) != 0x02014b50u) throw std::runtime_error("npz: bad central directory");
        const uint16_t method = rd16(cd.data() + p + 10);
        uint64_t csize = rd32(cd.data() + p + 20), usize = rd32(cd.data() + p + 24);
        const uint16_t nlen = rd16(cd.data() + p + 28), xlen = rd16(cd.data() + p + 30), clen = rd16(cd.data() + p + 32);
        uint64_t lho = rd32(cd.data() + p + 42);
        if (p + 46 + (size_t)nlen + xlen + clen > cd.size()) throw std::runtime_error("npz: bad central directory");
        std::string name(reinterpret_cast<const char *>(cd.data() + p + 46), nlen);
        apply_zip64_extra(cd.data() + p + 46 + nlen, xlen, usize, csize, lho);
        p += 46 + (size_t)nlen + xlen + clen;

        uint8_t lh[30];
        if (lho > fsize || fsize - lho < sizeof(lh)) throw std::runtime_error("npz: local header lies outside the file for " + name);
        f.read_at(lho, lh, sizeof(lh));
        if (rd32(lh) != 0x04034b50u) throw std::runtime_error("npz: bad local header for " + name);
        const uint64_t data_off = lho + 30 + rd16(lh + 26) + rd16(lh + 28);
        // sizes come from the file: a member cannot be longer than the file, and deflate expands by at most 1032:1
        if (data_off > fsize || csize > fsize - data_off) throw std::runtime_error("npz: member data lies outside the file for " + name);
        if ((method == 0 && usize != csize) || (method == 8 && usize / 1032 > csize + 1))
            throw std::runtime_error("npz: implausible member size for " + name);

        std::vector<uint8_t> raw(usize);
        if (method == 0) {
            f.read_at(data_off, raw.data(), usize);
        } else if (method == 8) {
            std::vector<uint8_t> comp(csize);
            f.read_at(data_off, comp.data(), csize);
            z_stream zs;
            std::memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) throw std::runtime_error("npz: inflateInit2 failed");
            // zlib counts in 32-bit uInt: feed both sides in bounded slices
            uint64_t in_done = 0, out_done = 0;
            for (;;) {
                if (zs.avail_in == 0 && in_done < csize) {
                    const uint64_t n = std::min<uint64_t>(csize - in_done, 1u << 30);
                    zs.next_in = comp.data() + in_done;
                    zs.avail_in = (uInt)n;
                    in_done += n;
                }
                if (zs.avail_out == 0 && out_done < usize) {
                    const uint64_t n = std::min<uint64_t>(usize - out_done, 1u << 30);
                    zs.next_out = raw.data() + out_done;
                    zs.avail_out = (uInt)n;
                    out_done += n;
                }
                const int rc = inflate(&zs, Z_NO_FLUSH);
                if (rc == Z_STREAM_END) break;
                if (rc != Z_OK) {
                    inflateEnd(&zs);
                    throw std::runtime_error("npz: inflate failed for " + name);
                }
                if (zs.avail_out == 0 && out_done >= usize) break;  // output complete
            }
            inflateEnd(&zs);
        } else {
            throw std::runtime_error("npz: unsupported compression method for " + name);
        }
        if (name.size() > 4 && name.compare(name.size() - 4, 4, ".npy") == 0) name.resize(name.size() - 4);
        out.emplace(std::move(name), parse_npy(raw.data(), raw.size()));
    }
    return out;
}

// ------------------------------------------------------------------ writer

Writer::Writer(const std::string &path) : fp_(std::fopen(path.c_str(), "wb")) {
    if (!fp_) throw std::runtime_error("npz: cannot create " + path);
}

Writer::~Writer() {
    if (fp_) {
        try { close(); } catch (...) {}
    }
}

void Writer::put(const void *p, size_t n) {
    if (std::fwrite(p, 1, n, fp_) != n) throw std::runtime_error("npz: write failed");
    pos_ += n;
}

static void le16(std::vector<uint8_t> &v, uint16_t x) { v.push_back(x & 0xff); v.push_back(x >> 8); }
static void le32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((x >> (8 * i)) & 0xff); }
static void le64(std::vector<uint8_t> &v, uint64_t x) { for (int i = 0; i < 8; ++i) v.push_back((x >> (8 * i)) & 0xff); }

void Writer::add(const std::string &name, const std::string &descr, const std::vector<size_t> &shape,
                 const void *data, size_t nbytes) {
    // .npy v1 header, padded so that the payload starts 64-byte aligned
    std::string shp = "(";
    for (size_t i = 0; i < shape.size(); ++i) {
        shp += std::to_string(shape[i]);
        if (shape.size() == 1) shp += ",";
        else if (i + 1 < shape.size()) shp += ", ";
    }
    shp += ")";
    std::string dict = "{'descr': '" + descr + "', 'fortran_order': False, 'shape': " + shp + ", }";
    size_t total = 10 + dict.size() + 1;
    const size_t pad = (64 - total % 64) % 64;
    dict.append(pad, ' ');
    dict.push_back('\n');
    std::vector<uint8_t> hdr = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    le16(hdr, (uint16_t)dict.size());
    hdr.insert(hdr.end(), dict.begin(), dict.end());

    const uint64_t size = hdr.size() + nbytes;
    uint32_t crc = (uint32_t)crc32(0L, hdr.data(), (uInt)hdr.size());
    const uint8_t *d = static_cast<const uint8_t *>(data);
    for (size_t off = 0; off < nbytes;) {
        const size_t n = std::min<size_t>(nbytes - off, 1u << 30);
        crc = (uint32_t)crc32(crc, d + off, (uInt)n);
        off += n;
    }
    const std::string fname = name + ".npy";
    const bool z64 = size >= 0xFFFFFFFFull || pos_ >= 0xFFFFFFFFull;
    std::vector<uint8_t> lh;
    le32(lh, 0x04034b50u);
    le16(lh, z64 ? 45 : 20);
    le16(lh, 0);
    le16(lh, 0);  // stored
    le16(lh, 0);
    le16(lh, 0x21);  // time, date (1980-01-01)
    le32(lh, crc);
    le32(lh, z64 ? 0xFFFFFFFFu : (uint32_t)size);
    le32(lh, z64 ? 0xFFFFFFFFu : (uint32_t)size);
    le16(lh, (uint16_t)fname.size());
    le16(lh, z64 ? 20 : 0);
    lh.insert(lh.end(), fname.begin(), fname.end());
    if (z64) {
        le16(lh, 0x0001);
        le16(lh, 16);
        le64(lh, size);
        le64(lh, size);
    }
    entries_.push_back({fname, crc, size, pos_});
    put(lh.data(), lh.size());
    put(hdr.data(), hdr.size());
    put(data, nbytes);
}

void Writer::add_unicode(const std::string &name, const std::string &ascii) {
    std::vector<uint32_t> u(ascii.begin(), ascii.end());
    add(name, "<U" + std::to_string(ascii.size()), {}, u.data(), u.size() * 4);
}

void Writer::close() {
    if (!fp_) return;
    const uint64_t cd_off = pos_;
    for (const Entry &e : entries_) {
        const bool z64 = e.size >= 0xFFFFFFFFull || e.offset >= 0xFFFFFFFFull;
        std::vector<uint8_t> c;
        le32(c, 0x02014b50u);
        le16(c, z64 ? 45 : 20);
        le16(c, z64 ? 45 : 20);
        le16(c, 0);
        le16(c, 0);
        le16(c, 0);
        le16(c, 0x21);
        le32(c, e.crc);
        le32(c, z64 ? 0xFFFFFFFFu : (uint32_t)e.size);
        le32(c, z64 ? 0xFFFFFFFFu : (uint32_t)e.size);
        le16(c, (uint16_t)e.name.size());
        le16(c, z64 ? 28 : 0);
        le16(c, 0);
        le16(c, 0);
        le16(c, 0);
        le32(c, 0);
        le32(c, z64 ? 0xFFFFFFFFu : (uint32_t)e.offset);
        c.insert(c.end(), e.name.begin(), e.name.end());
        if (z64) {
            le16(c, 0x0001);
            le16(c, 24);
            le64(c, e.size);
            le64(c, e.size);
            le64(c, e.offset);
        }
        put(c.data(), c.size());
    }
    const uint64_t cd_size = pos_ - cd_off;
    const bool z64 = entries_.size() >= 0xFFFF || cd_off >= 0xFFFFFFFFull || cd_size >= 0xFFFFFFFFull;
    if (z64) {
        std::vector<uint8_t> r;
        const uint64_t rec_off = pos_;
        le32(r, 0x06064b50u);
        le64(r, 44);
        le16(r, 45);
        le16(r, 45);
        le32(r, 0);
        le32(r, 0);
        le64(r, entries_.size());
        le64(r, entries_.size());
        le64(r, cd_size);
        le64(r, cd_off);
        le32(r, 0x07064b50u);
        le32(r, 0);
        le64(r, rec_off);
        le32(r, 1);
        put(r.data(), r.size());
    }
    std::vector<uint8_t> e;
    le32(e, 0x06054b50u);
    le16(e, 0);
    le16(e, 0);
    le16(e, (uint16_t)std::min<size_t>(entries_.size(), 0xFFFF));
    le16(e, (uint16_t)std::min<size_t>(entries_.size(), 0xFFFF));
    le32(e, (uint32_t)std::min<uint64_t>(cd_size, 0xFFFFFFFFull));
    le32(e, (uint32_t)std::min<uint64_t>(cd_off, 0xFFFFFFFFull));
    le16(e, 0);
    put(e.data(), e.size());
    std::fclose(fp_);
    fp_ = nullptr;
}

}  // namespace viewer::npz
