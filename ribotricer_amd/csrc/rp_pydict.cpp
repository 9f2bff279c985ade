// rp_pydict.cpp -- the reference's hand-over format read at C speed.
//
// merge_read_lengths (detect_orfs.py:54-83) returns  strand -> Counter{(chrom, pos): count};
// a caller that keeps the reference's pipeline and swaps in this package's export_orf_coverages
// hands that mapping over.  Turning 10 M dict entries into columns costs
// ~1.4 s of Python iteration; here one PyDict_Next loop fills the columns (CPython API, called
// through ctypes.PyDLL with the GIL held).  Host-side format conversion only: no scoring.
#include <Python.h>

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

extern "C" {

// One Counter -> columns.  `names` (a list) receives the chromosome names in order of first
// appearance ACROSS calls (pass the same list for every strand); chrom[i] indexes it.
// Returns the number of entries written, -1 on a malformed key / value (Python error set),
// -2 when `capacity` is too small.
long long rp_counter_columns(PyObject *table, PyObject *names, int32_t *chrom, int64_t *pos, int64_t *count,
                             long long capacity)
{
    if (!PyDict_Check(table) || !PyList_Check(names)) {
        PyErr_SetString(PyExc_TypeError, "rp_counter_columns needs a dict and a list");
        return -1;
    }
    // known names: by object identity first (the reference reuses one str per chromosome), then by value
    std::unordered_map<PyObject *, int32_t> by_object;
    std::unordered_map<std::string, int32_t> by_value;
    const Py_ssize_t n_known = PyList_GET_SIZE(names);
    for (Py_ssize_t k = 0; k < n_known; ++k) {
        PyObject *s = PyList_GET_ITEM(names, k);
        Py_ssize_t len = 0;
        const char *utf8 = PyUnicode_Check(s) ? PyUnicode_AsUTF8AndSize(s, &len) : nullptr;
        if (!utf8) return -1;
        by_value.emplace(std::string(utf8, (size_t)len), (int32_t)k);
    }
    Py_ssize_t at = 0;
    PyObject *key, *value, *last_object = nullptr;
    int32_t last_code = 0;
    long long n = 0;
    while (PyDict_Next(table, &at, &key, &value)) {
        if (n >= capacity) return -2;
        if (!PyTuple_Check(key) || PyTuple_GET_SIZE(key) != 2) {
            PyErr_SetString(PyExc_TypeError, "alignment keys must be (chrom, pos) tuples");
            return -1;
        }
        PyObject *c = PyTuple_GET_ITEM(key, 0);
        int32_t code;
        std::unordered_map<PyObject *, int32_t>::iterator hit;
        if (c == last_object) {  // (sorted BAMs: long runs of one chromosome)
            code = last_code;
        } else if ((hit = by_object.find(c)) != by_object.end()) {
            code = hit->second;
        } else {
            Py_ssize_t len = 0;
            const char *utf8 = PyUnicode_Check(c) ? PyUnicode_AsUTF8AndSize(c, &len) : nullptr;
            if (!utf8) {
                if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "chromosome names must be str");
                return -1;
            }
            std::string name(utf8, (size_t)len);
            auto v = by_value.find(name);
            if (v == by_value.end()) {
                code = (int32_t)PyList_GET_SIZE(names);
                if (PyList_Append(names, c) != 0) return -1;
                by_value.emplace(std::move(name), code);
            } else {
                code = v->second;
            }
            by_object.emplace(c, code);
        }
        last_object = c;
        last_code = code;
        const long long p = PyLong_AsLongLong(PyTuple_GET_ITEM(key, 1));
        if (p == -1 && PyErr_Occurred()) return -1;
        const long long v = PyLong_AsLongLong(value);
        if (v == -1 && PyErr_Occurred()) return -1;
        chrom[n] = code;
        pos[n] = p;
        count[n] = v;
        ++n;
    }
    return n;
}

}  // extern "C"
