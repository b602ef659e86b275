"""The file side of JRA55PrescribedAtmosphere / Radiation / Land, as far as this image allows.

Reference surface being mirrored:
  * `RepeatYearJRA55()` / `MultiYearJRA55()` dataset selectors and the keywords `dir, dataset, start_date, end_date,
    time_indices_in_memory (backend_size), prefetch` of `JRA55PrescribedAtmosphere(arch; …)` —
    /root/reference/src/OMIPConfigurations/atmosphere.jl:13-29;
  * the eleven yearly variable files `tas huss psl uas vas rlds rsds prra prsn friver licalvf` —
    jra55_data_staging.jl:8; 15–25 GB per forcing year (:134); 3-hourly snapshots on the 640 × 320 TL319 grid
    (launch.sh:86-87).

What is here: the dataset objects, the `start_date` / `end_date` → snapshot-counter mapping (repeat-year wrap, multi-year
clamping, leap days), and a snapshot PROVIDER — `provider(n) -> {variable: float32[320, 640]}` — for
models.JRA55PrescribedAtmosphere's sliding HBM window (cf_window_*), reading RAW little-endian Float32 planes with
`np.memmap`, NetCDF CLASSIC files (`nccopy -k classic tas_1990.nc4 tas_1990.nc`) through scipy.io.netcdf_file
(ClassicNetCDFFiles), or — round 6 — the distributed NetCDF-4 files themselves through a pure-Python reader of the HDF5 subset
they use (NetCDF4Files, coflux/hdf5_subset.py: chunked `[time, lat, lon]` variables, shuffle + deflate through zlib; written
from the format specification and tested against an independent writer of it, never against a file made by libhdf5 — this image
has no HDF5 library).  Where that reader refuses a file, the one-line conversion to raw planes a maintainer runs once per
yearly file, anywhere netCDF4 exists, is

    python -c "import netCDF4, sys; f, v = sys.argv[1:]; netCDF4.Dataset(f)[v][:].astype('<f4').tofile(f[:-3] + '.f32')" tas_1990.nc tas

(or `ncdump -v tas -p 9,9 tas_1990.nc` piped through any text → binary filter),

i.e. `<dir>/<var>_<year>.f32` = the variable's [time, lat, lon] array as C-ordered little-endian Float32, nothing else.
"""
import datetime as _dt
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import abi

JRA55_SHORTNAMES = ("tas", "huss", "psl", "uas", "vas", "rlds", "rsds", "prra", "prsn", "friver", "licalvf")  # jra55_data_staging.jl:8
# cf_atmos_source takes the nine atmosphere variables under these names and in this order (abi.JRA55_VARIABLES); the land pair
LAND_VARIABLES = ("friver", "licalvf")
assert JRA55_SHORTNAMES == tuple(abi.JRA55_VARIABLES) + LAND_VARIABLES
SNAPSHOT_INTERVAL = 3 * 3600.0   # seconds
NX, NY = 640, 320


def _snapshots_in_year(year, leap_days):
    days = 366 if (leap_days and (year % 4 == 0 and (year % 100 != 0 or year % 400 == 0))) else 365
    return days * 8


@dataclass(frozen=True)
class RepeatYearJRA55:
    """RepeatYearJRA55(): one forcing year (JRA55-do repeat-year forcing, May 1990 – April 1991 in the published product;
    here: whichever year the files hold) applied cyclically — time indices wrap at the end of the record."""
    year: int = 1990
    leap_days: bool = False
    cyclic = True

    def years(self, start_date=None, end_date=None):
        return (self.year,)


@dataclass(frozen=True)
class MultiYearJRA55:
    """MultiYearJRA55(): the interannual record 1958–…; time indices are clamped at the ends of [start_date, end_date]."""
    first_year: int = 1958
    last_year: int = 2023
    leap_days: bool = True
    cyclic = False

    def years(self, start_date=None, end_date=None):
        y0 = start_date.year if start_date else self.first_year
        y1 = end_date.year if end_date else self.last_year
        if y0 < self.first_year or y1 > self.last_year or y1 < y0:
            raise ValueError(f"MultiYearJRA55 covers {self.first_year}–{self.last_year}; asked for {y0}–{y1}")
        return tuple(range(y0, y1 + 1))


class SnapshotCalendar:
    """start_date / end_date → the run of 3-hourly snapshot records, and model time [s since start_date] → record index.

    Record r of the calendar is (year, k): snapshot k of that year's files.  `total` records; `record_of(n)` maps a
    snapshot COUNTER n (monotone, as models.JRA55PrescribedAtmosphere's window uses it) to (year, k): cyclic for a
    repeat year, clamped for a multi-year record."""

    def __init__(self, dataset, start_date=None, end_date=None):
        self.dataset = dataset
        years = dataset.years(start_date, end_date)
        start_date = start_date or _dt.datetime(years[0], 1, 1)
        end_date = end_date or _dt.datetime(years[-1] + 1, 1, 1)
        if end_date <= start_date:
            raise ValueError(f"end_date {end_date} is not after start_date {start_date}")
        self.start_date, self.end_date = start_date, end_date
        self.records = []
        for y in years:
            n = _snapshots_in_year(y, dataset.leap_days)
            t0 = _dt.datetime(y, 1, 1)
            for k in range(n):
                t = t0 + _dt.timedelta(seconds=k * SNAPSHOT_INTERVAL)
                if isinstance(dataset, RepeatYearJRA55) or (start_date <= t < end_date):
                    self.records.append((y, k))
        if isinstance(dataset, RepeatYearJRA55):
            # the repeat year starts at start_date's position within the year and wraps
            first = int(((start_date - _dt.datetime(start_date.year, 1, 1)).total_seconds()) // SNAPSHOT_INTERVAL) % len(self.records)
            self.records = self.records[first:] + self.records[:first]
        self.total = len(self.records)
        if self.total < 2:
            raise ValueError("fewer than two snapshots between start_date and end_date")

    def record_of(self, n):
        n = n % self.total if self.dataset.cyclic else min(max(n, 0), self.total - 1)
        return self.records[n]

    def time_of(self, n):
        """Model time [s] of snapshot counter n (n = 0 is the first snapshot at or after start_date)."""
        return n * SNAPSHOT_INTERVAL


class RawPlaneFiles:
    """`<dir>/<shortname>_<year>.f32`: [time, 320, 640] little-endian Float32, memory-mapped per (variable, year)."""

    def __init__(self, directory, shortnames=JRA55_SHORTNAMES, shape=(NY, NX)):
        self.dir, self.shape = directory, shape
        self.shortnames = tuple(shortnames)
        self._maps = {}

    def path(self, shortname, year):
        return os.path.join(self.dir, f"{shortname}_{year}.f32")

    def plane(self, shortname, year, k):
        key = (shortname, year)
        m = self._maps.get(key)
        if m is None:
            p = self.path(shortname, year)
            if not os.path.exists(p):
                raise FileNotFoundError(f"{p}: JRA55 plane file missing (see coflux/jra55.py for the one-line NetCDF → f32 conversion)")
            n = os.path.getsize(p) // (4 * self.shape[0] * self.shape[1])
            m = self._maps[key] = np.memmap(p, dtype="<f4", mode="r", shape=(n,) + tuple(self.shape))
        if not 0 <= k < m.shape[0]:
            raise IndexError(f"{self.path(shortname, year)} holds {m.shape[0]} snapshots, asked for {k}")
        return m[k]


class ClassicNetCDFFiles:
    """`<dir>/<shortname>_<year>.nc` in the NetCDF CLASSIC format (CDF-1/2/5 — what `nccopy -k classic` or `-k cdf5` writes
    from the distributed NetCDF-4 files), read with scipy.io.netcdf_file (memory-mapped; this image has no HDF5 library, so
    NetCDF-4 itself stays out of reach).  The variable is looked up by shortname, falling back to the file's only
    3-D variable; `scale_factor` / `add_offset` are applied, the fill value is passed through.  Same `plane()` contract
    as RawPlaneFiles: Float32 [320, 640]."""

    def __init__(self, directory, shortnames=JRA55_SHORTNAMES, shape=(NY, NX)):
        self.dir, self.shape = directory, tuple(shape)
        self.shortnames = tuple(shortnames)
        self._vars = {}

    def path(self, shortname, year):
        return os.path.join(self.dir, f"{shortname}_{year}.nc")

    def _variable(self, shortname, year):
        key = (shortname, year)
        v = self._vars.get(key)
        if v is None:
            from scipy.io import netcdf_file
            p = self.path(shortname, year)
            if not os.path.exists(p):
                raise FileNotFoundError(f"{p}: JRA55 NetCDF (classic) file missing")
            f = netcdf_file(p, "r", mmap=True)
            var = f.variables.get(shortname)
            if var is None:
                cubes = [x for x in f.variables.values() if len(x.shape) == 3]
                if len(cubes) != 1:
                    raise KeyError(f"{p}: no variable '{shortname}' and {len(cubes)} three-dimensional variables to choose from")
                var = cubes[0]
            if tuple(var.shape[1:]) != self.shape:
                raise ValueError(f"{p}: variable of shape {var.shape}, expected [time, {self.shape[0]}, {self.shape[1]}]")
            v = self._vars[key] = (f, var, float(getattr(var, "scale_factor", 1.0)), float(getattr(var, "add_offset", 0.0)))
        return v

    def plane(self, shortname, year, k):
        _, var, scale, offset = self._variable(shortname, year)
        if not 0 <= k < var.shape[0]:
            raise IndexError(f"{self.path(shortname, year)} holds {var.shape[0]} snapshots, asked for {k}")
        a = np.asarray(var[k], dtype=np.float32)   # big-endian on disk → native Float32
        if scale != 1.0 or offset != 0.0:
            a = (a * np.float32(scale) + np.float32(offset)).astype(np.float32)
        return a


class NetCDF4Files:
    """`<dir>/<shortname>_<year>.nc` (or `.nc4`) in the NetCDF-4 format the JRA55-do files are distributed in
    (jra55_data_staging.jl:8,134), read with coflux/hdf5_subset.py — no HDF5 library.  One time level per `plane()` call: with
    the files' one-level chunks that is one chunk located through the chunk B-tree, inflated and un-shuffled (≈ 1 ms for
    640×320 Float32).  `scale_factor` / `add_offset` are applied where the variable carries them.  Same contract as
    RawPlaneFiles: Float32 [320, 640]."""

    def __init__(self, directory, shortnames=JRA55_SHORTNAMES, shape=(NY, NX)):
        self.dir, self.shape = directory, tuple(shape)
        self.shortnames = tuple(shortnames)
        self._vars = {}

    def path(self, shortname, year):
        for ext in (".nc", ".nc4"):
            p = os.path.join(self.dir, f"{shortname}_{year}{ext}")
            if os.path.exists(p):
                return p
        return os.path.join(self.dir, f"{shortname}_{year}.nc")

    def _variable(self, shortname, year):
        key = (shortname, year)
        v = self._vars.get(key)
        if v is None:
            from . import hdf5_subset
            p = self.path(shortname, year)
            if not os.path.exists(p):
                raise FileNotFoundError(f"{p}: JRA55 NetCDF-4 file missing")
            f, ds = hdf5_subset.open_variable(p, shortname)
            if len(ds.shape) != 3 or tuple(ds.shape[1:]) != self.shape:
                raise ValueError(f"{p}: variable of shape {ds.shape}, expected [time, {self.shape[0]}, {self.shape[1]}]")
            v = self._vars[key] = (f, ds, float(ds.attrs.get("scale_factor", 1.0)), float(ds.attrs.get("add_offset", 0.0)))
        return v

    def plane(self, shortname, year, k):
        _, ds, scale, offset = self._variable(shortname, year)
        if not 0 <= k < ds.shape[0]:
            raise IndexError(f"{self.path(shortname, year)} holds {ds.shape[0]} snapshots, asked for {k}")
        a = ds.read_leading(k).astype(np.float32)
        if scale != 1.0 or offset != 0.0:
            a = (a * np.float32(scale) + np.float32(offset)).astype(np.float32)
        return a


def _is_hdf5(path):
    try:
        with open(path, "rb") as fh:
            return fh.read(8) == b"\x89HDF\r\n\x1a\n"
    except OSError:
        return False


def plane_files(directory):
    """RawPlaneFiles, ClassicNetCDFFiles or NetCDF4Files, by what the directory holds (`*.f32` wins; a `.nc` file is told apart
    by its first eight bytes: the HDF5 signature means NetCDF-4)."""
    names = os.listdir(directory) if os.path.isdir(directory) else []
    nc = sorted(n for n in names if n.endswith((".nc", ".nc4")))
    if any(n.endswith(".f32") for n in names) or not nc:
        return RawPlaneFiles(directory)
    if _is_hdf5(os.path.join(directory, nc[0])):
        return NetCDF4Files(directory)
    return ClassicNetCDFFiles(directory)


def atmosphere_provider(directory, calendar, files=None):
    """provider(n) for models.JRA55PrescribedAtmosphere: snapshot counter → {cf_atmos_source variable: float32[320, 640]}."""
    files = files or plane_files(directory)

    def provider(n):
        year, k = calendar.record_of(n)
        return {var: files.plane(var, year, k) for var in abi.JRA55_VARIABLES}
    return provider


def land_provider(directory, calendar, files=None):
    """provider(n) for models.JRA55PrescribedLand's sliding window: snapshot counter → {friver, licalvf: float32[320, 640]}."""
    files = files or plane_files(directory)

    def provider(n):
        year, k = calendar.record_of(n)
        return {var: files.plane(var, year, k) for var in LAND_VARIABLES}
    return provider


def land_snapshots(directory, calendar, first=0, count=2, files=None):
    """{friver, licalvf}: float32[count, 320, 640] for JRA55PrescribedLand's in-memory window."""
    files = files or plane_files(directory)
    out = {}
    for var in LAND_VARIABLES:
        out[var] = np.stack([files.plane(var, *calendar.record_of(first + n)) for n in range(count)]).astype(np.float32)
    return out


def write_raw_year(directory, year, snapshots, land=None):
    """The inverse of the readers (tests, and the conversion step when the NetCDF side is read elsewhere):
    snapshots = {JRA55 shortname: float32[n, 320, 640]} → `<dir>/<shortname>_<year>.f32`."""
    os.makedirs(directory, exist_ok=True)
    for var in abi.JRA55_VARIABLES:
        np.ascontiguousarray(snapshots[var], dtype="<f4").tofile(os.path.join(directory, f"{var}_{year}.f32"))
    for var in LAND_VARIABLES:
        if land is not None and var in land:
            np.ascontiguousarray(land[var], dtype="<f4").tofile(os.path.join(directory, f"{var}_{year}.f32"))
