"""
`ImInfo` for the hot path without tifffile / ome_types: the duck type Filter and Label use
(nellie/im_info/verifier.py:698-1070 -- `no_z`, `no_t`, `shape`, `axes`, `dim_res`, `im_path`, `im`,
`pipeline_paths`, `get_memmap`, `allocate_memory`, `create_output_path`, `remove_intermediates`) on top of
nellie_amd.im_info.ome_tiff, with the reference's directory layout and "detailed" file naming
(verifier.py:574-618): `<dir>/nellie_output/nellie_necessities/<name>-<axes>-<dims>-ch<c>[-t<a>_to_<b>]-<stage>.ome.tif`.

Inside a Nellie installation use Nellie's own FileInfo / ImInfo (metadata sniffing, ND2, channel and time
selection live there and are out of scope here); this class covers arrays and already-canonical TIFFs.
"""
from __future__ import annotations

import os

import numpy as np

from nellie_amd.im_info import ome_tiff

_PIPELINE = [("im_preprocessed", ".ome.tif", True), ("im_instance_label", ".ome.tif", True), ("im_skel", ".ome.tif", True),
             ("im_skel_relabelled", ".ome.tif", True), ("im_pixel_class", ".ome.tif", True), ("im_marker", ".ome.tif", True),
             ("im_distance", ".ome.tif", True), ("im_border", ".ome.tif", True), ("flow_vector_array", ".npy", True),
             ("voxel_matches", ".npy", True), ("im_branch_label_reassigned", ".ome.tif", True),
             ("im_obj_label_reassigned", ".ome.tif", True), ("features_voxels", ".csv", False),
             ("features_nodes", ".csv", False), ("features_branches", ".csv", False), ("features_organelles", ".csv", False),
             ("features_image", ".csv", False), ("adjacency_maps", ".pkl", True)]


def _canonical(data, axes):
    """verifier.py:889-929: T first (added when missing), singleton Z squeezed, order T[Z]YX."""
    axes_list = list(axes)
    if "T" not in axes_list:
        data = data[np.newaxis, ...]
        axes_list = ["T"] + axes_list
    else:
        ti = axes_list.index("T")
        if ti != 0:
            data = np.moveaxis(data, ti, 0)
            axes_list = ["T"] + [a for i, a in enumerate(axes_list) if i != ti]
    if "Z" in axes_list:
        zi = axes_list.index("Z")
        if data.shape[zi] == 1:
            data = np.squeeze(data, axis=zi)
            axes_list.pop(zi)
    extra = [a for a in axes_list if a not in "TZYX"]
    if extra:
        raise ValueError(f"Unsupported axes found: {extra}")
    if "Y" not in axes_list or "X" not in axes_list:
        raise ValueError("Axes must include both Y and X")
    target = ["T"] + (["Z"] if "Z" in axes_list else []) + ["Y", "X"]
    if axes_list != target:
        data = np.transpose(data, [axes_list.index(a) for a in target])
        axes_list = target
    return data, "".join(axes_list)


def detailed_output_name(name, axes, dim_res, ch, t_start, t_end):
    """Nellie's "detailed" output name (verifier.py:596-613): <name>-<axes>-<axis><resolution>_...-ch<c>[-t<a>_to_<b>], the
    resolutions rounded to four decimals with '.' written as 'p', axes without a resolution entry (C) left out, the time
    range present when the source has a T axis."""
    parts = []
    for axis in axes:
        if axis in dim_res:
            value = dim_res[axis]
            parts.append(axis + ("None" if value is None else str(round(value, 4)).replace(".", "p")))
    t_range = f"-t{t_start}_to_{t_end}" if "T" in axes else ""
    return f"{name}-{axes}-{'_'.join(parts)}-ch{ch}{t_range}"


class FileInfo:
    """The part of Nellie's FileInfo (verifier.py:17-700) that `run(file_info)` and `ImInfo(file_info)` consume: where the
    image lives, its axes, resolutions, channel and time range.  Sources: OME-TIFFs as this package (and Nellie) write
    them, and .npy arrays; metadata sniffing of arbitrary TIFF / ND2 files stays with Nellie's own class."""

    def __init__(self, filepath, output_dir=None, output_naming="detailed"):
        self.filepath = os.fspath(filepath)
        self.output_dir = output_dir
        if output_naming not in ("detailed", "stable"):
            raise ValueError(f"Unsupported output naming strategy '{output_naming}'")
        self.output_naming = output_naming
        # verifier.py:126: splitext of the basename and nothing more ("foo.ome.tif" -> "foo.ome"): Nellie's later stages rebuild
        # the paths of these files from their own FileInfo, so the names have to agree character for character
        self.filename_no_ext = os.path.splitext(os.path.basename(self.filepath))[0]
        self.axes = None
        self.shape = None
        self.dim_res = {"X": None, "Y": None, "Z": None, "T": None}
        self.ch = 0
        self.t_start = self.t_end = None
        self.good_dims = self.good_axes = False

    def find_metadata(self):
        if self.filepath.lower().endswith(".npy"):
            data = np.load(self.filepath, mmap_mode="r")
            self.axes = {2: "YX", 3: "ZYX", 4: "TZYX"}[data.ndim]
        else:
            data, lay = ome_tiff.memmap(self.filepath, mode="r")
            self.axes = lay.axes
            self.dim_res.update(lay.dim_res)
        self.shape = tuple(data.shape)
        return self

    def _check_time_range(self):
        """verifier.py:393-408, 526-539: an out-of-range selection is an error, not a shorter stack."""
        if self.axes is None or self.shape is None or "T" not in self.axes or self.t_start is None or self.t_end is None:
            return
        max_t = self.shape[self.axes.index("T")] - 1
        if self.t_start < 0 or self.t_end < 0:
            raise ValueError("Temporal range must be >= 0")
        if self.t_start > self.t_end:
            raise ValueError("Start frame must be <= end frame")
        if self.t_start > max_t or self.t_end > max_t:
            raise ValueError("Temporal range out of bounds")

    def load_metadata(self):
        if self.axes is None:
            self.find_metadata()
        if "T" in self.axes:
            if self.t_start is None:
                self.t_start = 0
            if self.t_end is None:
                self.t_end = self.shape[self.axes.index("T")] - 1
        self._check_time_range()
        self.good_axes = all(a in "TZYX" for a in self.axes) and "X" in self.axes and "Y" in self.axes
        self.good_dims = all(self.dim_res.get(a) is not None for a in self.axes if a in self.dim_res)
        return self

    def change_axes(self, new_axes):
        if self.shape is not None and len(new_axes) != len(self.shape):
            raise ValueError("New axes must have the same length as the existing shape")
        self.axes = new_axes
        return self.load_metadata()

    def change_dim_res(self, dim, new_size):
        if dim not in self.dim_res:
            raise ValueError("Invalid dimension")
        self.dim_res[dim] = new_size
        return self.load_metadata()

    def change_selected_channel(self, ch):
        self.ch = int(ch)

    def select_temporal_range(self, start=0, end=None):
        """verifier.py:475-506, same exceptions."""
        if self.axes is None:
            self.find_metadata()
        if "T" not in self.axes:
            raise KeyError("No time dimension to select")
        if start < 0:
            raise IndexError("Start frame must be >= 0")
        max_t = self.shape[self.axes.index("T")] - 1
        if end is None:
            end = max_t
        if end < 0:
            raise IndexError("End frame must be >= 0")
        if start > end:
            raise ValueError("Start frame must be <= end frame")
        if start > max_t or end > max_t:
            raise IndexError("Temporal range out of bounds")
        self.t_start, self.t_end = int(start), int(end)


class ImInfo:
    def __init__(self, source, dim_res=None, axes=None, output_dir=None, name=None, ch=0):
        """
        source : numpy array, or path to a .npy file or to an uncompressed contiguous TIFF / OME-TIFF.
        dim_res: {'X','Y','Z','T'} in um / s (taken from the OME metadata of a TIFF source when omitted).
        axes   : axes of an array / .npy source, e.g. 'ZYX' or 'TZYX' (default by rank: YX, ZYX, TZYX).
        """
        lay = None
        t_range = None
        output_naming = "detailed"
        if isinstance(source, FileInfo):                 # ImInfo(file_info), as nellie.run.run builds it (run.py:49)
            fi = source if source.axes is not None else source.load_metadata()
            fi.load_metadata()
            dim_res = dict(fi.dim_res) if dim_res is None else dim_res
            axes, ch, name = axes or fi.axes, fi.ch, name or fi.filename_no_ext
            output_dir = output_dir or fi.output_dir
            t_range = (fi.t_start, fi.t_end) if fi.t_start is not None else None
            output_naming = fi.output_naming
            source = fi.filepath
        if isinstance(source, (str, os.PathLike)):
            src_path = os.fspath(source)
            base_dir = os.path.dirname(os.path.abspath(src_path))
            name = name or os.path.splitext(os.path.basename(src_path))[0]
            if src_path.lower().endswith(".npy"):
                data = np.load(src_path, mmap_mode="r")
            else:
                data, lay = ome_tiff.memmap(src_path, mode="r")
                axes = axes or lay.axes
                if dim_res is None:
                    dim_res = lay.dim_res
        else:
            data = np.asarray(source)
            base_dir = os.getcwd()
            name = name or "array"
        if axes is None:
            axes = {2: "YX", 3: "ZYX", 4: "TZYX"}[data.ndim]
        if len(axes) != data.ndim:
            raise ValueError("Data dimensions do not match axes")
        self.dim_res = {"X": None, "Y": None, "Z": None, "T": None}
        self.dim_res.update(dim_res or {})
        src_axes = axes
        data, self.axes = _canonical(data, axes)
        t_first, t_last = 0, data.shape[0] - 1
        if t_range is not None and "T" in src_axes:       # the selected time range (verifier.py:640-660)
            t_first, t_last = t_range
            if not (0 <= t_first <= t_last <= data.shape[0] - 1):
                raise ValueError("Temporal range out of bounds")      # the reference's np.take raises here (verifier.py:640-660)
            data = data[t_first:t_last + 1]
        self.new_axes = self.axes
        self.shape = data.shape
        self.ch = ch
        self.output_dir = os.path.join(output_dir or base_dir, "nellie_output")
        self.nellie_necessities_dir = os.path.join(self.output_dir, "nellie_necessities")
        os.makedirs(self.nellie_necessities_dir, exist_ok=True)
        # "detailed" naming (verifier.py:596-613), built from the SOURCE axes like FileInfo does
        # "stable" naming is the bare file name (verifier.py:597-598)
        output_name = name if output_naming == "stable" else detailed_output_name(name, src_axes, self.dim_res, ch, t_first, t_last)
        self.user_output_path_no_ext = os.path.join(self.output_dir, output_name)
        self.nellie_necessities_output_path_no_ext = os.path.join(self.nellie_necessities_dir, output_name)
        self.im_path = self.nellie_necessities_output_path_no_ext + ".ome.tif"
        # the re-saved, canonical input (verifier.py:620-695); always T[Z]YX on disk here
        shape4 = self._shape4()
        # A file source is re-saved once (the reference keys its cache on the source file, verifier.py:620-628).  An in-memory
        # array or a .npy has no such identity -- a second ImInfo built from DIFFERENT pixels of the same shape would find
        # the first one's canonical copy under the same name and the stages would silently process stale data -- so for
        # those the canonical input is always rewritten.
        # The detailed name carries the time range, so a canonical copy found under it is this selection's -- provided it has the
        # expected shape and dtype (a "stable" name does not say which frames it holds: rewritten unless the whole stack is taken).
        from_file = isinstance(source, (str, os.PathLike)) and not os.fspath(source).lower().endswith(".npy")
        if from_file and output_naming == "stable" and t_range is not None and "T" in src_axes:
            from_file = False
        reuse = False
        if from_file and os.path.exists(self.im_path):
            try:
                old, _ = ome_tiff.memmap(self.im_path, mode="r")
                reuse = int(np.prod(old.shape)) == int(np.prod(shape4)) and old.dtype == data.dtype
                del old
            except Exception:
                reuse = False
        if not reuse:
            ome_tiff.create(self.im_path, shape4, data.dtype, self.dim_res, "input", data=np.asarray(data).reshape(shape4))
        self.im = self.get_memmap(self.im_path)
        self.no_z = not ("Z" in self.axes and self.shape[self.axes.index("Z")] > 1)
        self.no_t = not ("T" in self.axes and self.shape[self.axes.index("T")] > 1)
        self.pipeline_paths = {}
        for stage, ext, for_nellie in _PIPELINE:
            self.create_output_path(stage, ext, for_nellie)

    def _shape4(self):
        return (self.shape[0], self.shape[1] if "Z" in self.axes else 1, self.shape[-2], self.shape[-1])

    def create_output_path(self, pipeline_path, ext=".ome.tif", for_nellie=True):
        base = self.nellie_necessities_output_path_no_ext if for_nellie else self.user_output_path_no_ext
        self.pipeline_paths[pipeline_path] = f"{base}-{pipeline_path}{ext}"
        return self.pipeline_paths[pipeline_path]

    def remove_intermediates(self):
        for path in list(self.pipeline_paths.values()) + [self.im_path]:
            if "csv" in path:
                continue
            if os.path.exists(path):
                os.remove(path)

    def get_memmap(self, file_path, read_mode="r+"):
        """verifier.py:967-990: memory map in this ImInfo's canonical axes."""
        mm, lay = ome_tiff.memmap(file_path, mode=read_mode)
        data, axes = _canonical(mm, lay.axes)
        if axes != self.axes:
            raise ValueError(f"Axes mismatch: file has {axes}, ImInfo expects {self.axes}")
        return data

    def allocate_memory(self, output_path, dtype="float", data=None, description="No description.",
                        return_memmap=False, read_mode="r+"):
        """verifier.py:992-1070: a zero-filled (or data-filled) OME-BigTIFF of this image's shape."""
        if data is not None:
            data = np.asarray(data)
            if data.ndim == len(self.axes) - 1:
                data = data[np.newaxis, ...]
            if data.shape != tuple(self.shape):
                raise ValueError("Data dimensions do not match axes")
            dtype = data.dtype
            data = data.reshape(self._shape4())
        ome_tiff.create(output_path, self._shape4(), np.dtype(dtype), self.dim_res, description, data=data)
        if return_memmap:
            return self.get_memmap(output_path, read_mode=read_mode)
