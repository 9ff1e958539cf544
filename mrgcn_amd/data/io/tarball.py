"""Reader (and writer) of the reference's dataset archive (mrgcn/data/io/tarball.py:13-330): a plain
tar whose members are
    <name>.npz                      scipy CSR as data / indices / indptr / shape (read back as float32, :151-157)
    <name>.npy | .pt | .pkl         numpy array | torch tensor | pickled Python object
    dict/<name>/<k1>/.../<leaf>.<ext>   nested dict, one member per leaf (an empty dict is a directory entry)
    list/<name>/<i>.<ext>           list, members sorted by path
    <name>/{indices,values,size}.pt a torch sparse COO tensor
`mkdataset.py:121-122` stores a dataset as A, F, Y, data, sample_map, class_map; `run.py:63-69` reads
them back with `get`.  Same constructor / `get` / `read` / `store` / `list_members` surface here, so that
tarballs built elsewhere with the reference feed this package unchanged."""
from __future__ import annotations

import io
import os
import pickle
import tarfile

import numpy as np
import scipy.sparse as sp
import torch


class Tarball:
    def __init__(self, path=None, mode="r", separator="/"):
        if path is None:
            raise ValueError("::No path supplied")
        self.separator = separator
        self.tar = tarfile.open(path, mode)
        self._content = self.read(path) if "r" in mode else {}

    # ---- reading ---------------------------------------------------------------------------------
    def _blob(self, member: str) -> io.BytesIO:
        return io.BytesIO(self.tar.extractfile(member).read())

    def _load(self, member: str):
        ext = os.path.splitext(member)[1]
        if ext == ".npz":
            z = np.load(self._blob(member))
            return sp.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=z["shape"], dtype=np.float32)
        if ext == ".npy":
            return np.load(self._blob(member), allow_pickle=True)
        if ext == ".pt":
            return torch.load(self._blob(member))
        return pickle.load(self._blob(member))

    def read(self, path=None):
        sep = self.separator
        out = {}
        infos = {m.name: m for m in self.tar.getmembers()}
        nested = sorted(n for n in infos if sep in n)
        sparse_pt = {}
        for name in nested:
            parts = name.split(sep)
            if parts[0] == "dict":
                if len(parts) < 2:
                    continue
                node = out.setdefault(parts[1], {})
                if infos[name].isdir():          # an empty (sub)dict
                    for k in parts[2:]:
                        node = node.setdefault(k, {})
                    continue
                for k in parts[2:-1]:
                    node = node.setdefault(k, {})
                if len(parts) > 2:
                    node[os.path.splitext(parts[-1])[0]] = self._load(name)
            elif parts[0] == "list":
                out.setdefault(parts[1], []).append(self._load(name))   # `nested` is sorted: list order
            elif parts[-1] in ("indices.pt", "values.pt", "size.pt") and len(parts) == 2:
                sparse_pt.setdefault(parts[0], {})[parts[-1]] = name
        for item, files in sparse_pt.items():
            if set(files) == {"indices.pt", "values.pt", "size.pt"}:
                out[item] = torch.sparse_coo_tensor(self._load(files["indices.pt"]), self._load(files["values.pt"]),
                                                    tuple(self._load(files["size.pt"])))
        for name, info in infos.items():
            if sep not in name and info.isfile():
                out[os.path.splitext(name)[0]] = self._load(name)
        return out

    def get(self, key):
        return self._content[key]

    def list_members(self):
        names = set()
        for n in self.tar.getnames():
            parts = n.split(self.separator)
            if len(parts) > 1:
                names.add(parts[1] if parts[0] in ("dict", "list") else parts[0])
            else:
                names.add(os.path.splitext(n)[0])
        return list(names)

    def __len__(self):
        return len(self._content)

    # ---- writing ---------------------------------------------------------------------------------
    def _add(self, payload: bytes, name: str):
        info = tarfile.TarInfo(name=name)
        info.size = len(payload)
        self.tar.addfile(tarinfo=info, fileobj=io.BytesIO(payload))

    def _store_leaf(self, obj, name: str):
        buf = io.BytesIO()
        if sp.issparse(obj) and obj.format == "csr":
            np.savez(buf, data=obj.data, indices=obj.indices, indptr=obj.indptr, shape=obj.shape)
            self._add(buf.getvalue(), name + ".npz")
        elif isinstance(obj, np.ndarray):
            np.save(buf, obj)
            self._add(buf.getvalue(), name + ".npy")
        elif isinstance(obj, torch.Tensor) and obj.layout is torch.sparse_coo:
            c = obj.coalesce()
            for part, t in (("indices", c.indices()), ("values", c.values()), ("size", c.size())):
                self._store_leaf(t if isinstance(t, torch.Tensor) else torch.Size(t), os.path.join(name, part))
        elif isinstance(obj, (torch.Tensor, torch.Size)):
            torch.save(obj, buf)
            self._add(buf.getvalue(), name + ".pt")
        else:
            pickle.dump(obj, buf, protocol=4)
            self._add(buf.getvalue(), name + ".pkl")

    def _store_dict(self, obj, name: str):
        if not isinstance(obj, dict):
            return self._store_leaf(obj, name)
        if not obj:
            info = tarfile.TarInfo(name)
            info.type = tarfile.DIRTYPE
            return self.tar.addfile(info)
        for k, v in obj.items():
            self._store_dict(v, os.path.join(name, k))

    def store(self, files, names):
        assert len(files) == len(names)
        for obj, name in zip(files, names):
            if isinstance(obj, dict):
                self._store_dict(obj, os.path.join("dict", name))
            elif isinstance(obj, list):
                for i, item in enumerate(obj):
                    self._store_leaf(item, os.path.join("list", name, str(i)))
            else:
                self._store_leaf(obj, name)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.tar.close()
