// Fixture / mesh I/O of the facade: the CSV dialect of the reference's data files and the loader of its mesh directories.
//   CSVReader<T>::parse_file   fdaPDE/utils/IO/csv_reader.h:75-117  (dense files: one header row, first column = row index, blanks and
//                              double quotes dropped from every token, NA / NaN / nan -> quiet NaN)
//   CSVReader<T>::parse_sparse_file   csv_reader.h:119-166 (the reference's parse_file<Eigen::Sparse>: 3-column files "index, row, col,
//                              value", 1-based ids; the matrix is max row x max col, entries given twice are summed as setFromTriplets does)
//   MeshLoader<M, N>           test/src/utils/mesh_loader.h:62-84     (points.csv, elements.csv, boundary.csv, edges.csv, neigh.csv;
//                              ids are 1-based in the files: elements - 1; edges / neigh: x > 0 ? x - 1 : -1)
// Written against the facade's own containers (single pass over the file, no second read for the size as the reference does).
#ifndef FDAPDE_AMD_IO_H
#define FDAPDE_AMD_IO_H

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "pde.h"

namespace fdapde {
namespace amd {

template <typename T> class CSVReader {
   public:
    CSVReader() = default;
    DMatrix<T> parse_file(const std::string& file) const {
        std::ifstream in(file, std::ios::binary);
        if (!in) throw std::runtime_error("CSVReader: cannot open " + file);
        std::string line, tok;
        if (!std::getline(in, line)) return DMatrix<T>();
        const int64_t cols = count_fields(line) - 1;   // the first column is the row index
        std::vector<T> values;
        int64_t rows = 0;
        while (std::getline(in, line)) {
            if (line.empty() || line == "\r") continue;
            int64_t field = 0;
            size_t pos = 0;
            while (pos <= line.size()) {
                const size_t end = std::min(line.find(',', pos), line.size());
                if (field > 0) {
                    if (field > cols) throw std::runtime_error("CSVReader: too many fields in a row of " + file);
                    tok.clear();
                    for (size_t k = pos; k < end; ++k) {
                        const char ch = line[k];
                        if (ch != ' ' && ch != '"' && ch != '\r') tok += ch;
                    }
                    values.push_back(convert(tok));
                }
                ++field, pos = end + 1;
            }
            if (field - 1 != cols) throw std::runtime_error("CSVReader: ragged row in " + file);
            ++rows;
        }
        DMatrix<T> m(rows, cols);
        for (int64_t i = 0; i < rows; ++i)
            for (int64_t j = 0; j < cols; ++j) m(i, j) = values[(size_t)(i * cols + j)];
        return m;
    }
    // sparse files: one header row with four fields, then one line "index, row, col, value" per entry (ids 1-based)
    SpMatrix<T> parse_sparse_file(const std::string& file) const {
        std::ifstream in(file, std::ios::binary);
        if (!in) throw std::runtime_error("CSVReader: cannot open " + file);
        std::string line;
        if (!std::getline(in, line) || count_fields(line) - 1 != 3) throw std::runtime_error(".csv file not in sparse 3-column format");
        struct Entry {
            int64_t row, col;
            T value;
        };
        std::vector<Entry> entries;
        int64_t n_rows = 0, n_cols = 0;
        std::string tok[4];
        while (std::getline(in, line)) {
            if (line.empty() || line == "\r") continue;
            int field = 0;
            size_t pos = 0;
            while (pos <= line.size() && field < 4) {
                const size_t end = std::min(line.find(',', pos), line.size());
                tok[field].clear();
                for (size_t k = pos; k < end; ++k)
                    if (line[k] != ' ' && line[k] != '"' && line[k] != '\r') tok[field] += line[k];
                ++field, pos = end + 1;
            }
            if (field != 4) throw std::runtime_error("CSVReader: a line of " + file + " does not have three data fields");
            const int64_t r = std::strtoll(tok[1].c_str(), nullptr, 10), c = std::strtoll(tok[2].c_str(), nullptr, 10);
            if (r < 1 || c < 1) throw std::runtime_error("CSVReader: ids of a sparse file are 1-based (" + file + ")");
            n_rows = std::max(n_rows, r), n_cols = std::max(n_cols, c);
            entries.push_back({r - 1, c - 1, convert(tok[3])});
        }
        // CSR with sorted columns; an entry that appears more than once is the sum of its occurrences, in file order
        std::stable_sort(entries.begin(), entries.end(), [](const Entry& a, const Entry& b) { return a.row != b.row ? a.row < b.row : a.col < b.col; });
        SpMatrix<T> m;
        m.n_rows = n_rows, m.n_cols = n_cols;
        m.rowptr.assign((size_t)n_rows + 1, 0);
        for (size_t k = 0; k < entries.size(); ++k) {
            if (k > 0 && entries[k].row == entries[k - 1].row && entries[k].col == entries[k - 1].col) {
                m.values.back() += entries[k].value;
                continue;
            }
            m.colidx.push_back((int32_t)entries[k].col), m.values.push_back(entries[k].value);
            ++m.rowptr[(size_t)entries[k].row + 1];
        }
        for (int64_t i = 0; i < n_rows; ++i) m.rowptr[(size_t)i + 1] += m.rowptr[(size_t)i];
        return m;
    }
   private:
    static int64_t count_fields(const std::string& line) {
        int64_t n = 1;
        for (char ch : line) n += ch == ',';
        return n;
    }
    static T convert(const std::string& tok) {
        if (tok == "NA" || tok == "NaN" || tok == "nan") return std::numeric_limits<T>::quiet_NaN();   // 0 for integral T, as in the reference
        if (tok.empty()) return T();
        return (T)std::strtod(tok.c_str(), nullptr);
    }
};

// loads <directory>/<mesh id>/{points, elements, boundary, edges, neigh}.csv; edges / neigh are optional (empty if absent)
template <int M, int N> struct MeshLoader {
    DMatrix<double> points_;
    DMatrix<int> elements_, edges_, boundary_, neighbors_;
    Triangulation<M, N> mesh;
    MeshLoader(const std::string& directory, const std::string& mesh_id) {
        const std::string base = directory + (directory.empty() || directory.back() == '/' ? "" : "/") + mesh_id + "/";
        CSVReader<double> dr;
        CSVReader<int> ir;
        points_ = dr.parse_file(base + "points.csv");
        elements_ = ir.parse_file(base + "elements.csv");
        for (int64_t i = 0; i < elements_.size(); ++i) elements_.data()[i] -= 1;   // realign indexes to 0
        boundary_ = ir.parse_file(base + "boundary.csv");
        auto optional = [&](const std::string& f) {
            std::ifstream probe(f);
            if (!probe) return DMatrix<int>();
            DMatrix<int> m = ir.parse_file(f);
            for (int64_t i = 0; i < m.size(); ++i) m.data()[i] = m.data()[i] > 0 ? m.data()[i] - 1 : -1;
            return m;
        };
        edges_ = optional(base + "edges.csv");
        neighbors_ = optional(base + "neigh.csv");
        mesh = Triangulation<M, N>(points_, elements_, boundary_);
    }
};

}   // namespace amd
}   // namespace fdapde

#endif
