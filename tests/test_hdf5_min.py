"""CPU: the minimal HDF5 reader (vae_captioning_amd/utils/hdf5_min.py) against an independent writer of the same structures
(tests/hdf5_writer.py): the `images (N, 224, 224, 3) uint8` file of preprocess.py:25-45 as utils/batch_gen.py:35-41,189 reads it.
FORMAT-SPEC status: neither h5py nor libhdf5 exists in this image; no file written by them has been read."""
import numpy as np
import pytest

from vae_captioning_amd.utils import hdf5_min
from vae_captioning_amd.utils.batch_gen import open_image_array

from . import hdf5_writer


def _images(n, seed=0):
    return np.random.default_rng(seed).integers(0, 256, size=(n, 224, 224, 3), dtype=np.uint8)


@pytest.mark.parametrize("kw", [dict(), dict(user_block=512), dict(two_level_btree=True), dict(continuation=True)],
                         ids=["plain", "user-block", "two-level-btree", "header-continuation"])
def test_reads_the_image_array_like_the_reference_does(tmp_path, kw):
    imgs = _images(7)
    other = np.arange(24, dtype="<f4").reshape(2, 3, 4)
    p = str(tmp_path / "train_val.hdf5")
    hdf5_writer.write(p, {"images": imgs, "aaa": other, "zz_last": np.arange(5, dtype=">i2")}, **kw)
    with hdf5_min.File(p) as f:
        assert f.keys() == ["aaa", "images", "zz_last"] and "images" in f and "nope" not in f
        d = f["images"]
        assert d.shape == (7, 224, 224, 3) and d.dtype == np.uint8 and len(d) == 7
        np.testing.assert_array_equal(d[3], imgs[3])
        idx = sorted([5, 0, 2])                       # utils/batch_gen.py:347-362: sorted indices, fancy indexing on the first axis
        np.testing.assert_array_equal(d[idx], imgs[idx])
        np.testing.assert_array_equal(d[1:4], imgs[1:4])
        np.testing.assert_array_equal(np.asarray(f["aaa"]), other)
        z = f["zz_last"]
        assert z.dtype == np.dtype(">i2")
        np.testing.assert_array_equal(np.asarray(z), np.arange(5))
        with pytest.raises(KeyError, match="no object 'missing'"):
            f["missing"]
    # the product's entry point: h5py absent -> hdf5_min
    np.testing.assert_array_equal(open_image_array(p)[[1, 6]], imgs[[1, 6]])


def test_a_data_set_that_was_never_written_reads_as_zeros(tmp_path):
    p = str(tmp_path / "empty.h5")
    hdf5_writer.write(p, {"images": ((3, 4, 4, 3), "u1")})
    d = hdf5_min.File(p)["images"]
    assert d.shape == (3, 4, 4, 3) and not np.asarray(d).any()


def test_refuses_what_it_does_not_read(tmp_path):
    p = str(tmp_path / "c.h5")
    for layout, word in (("chunked", "chunked"), ("compact", "compact")):
        hdf5_writer.write(p, {"images": _images(1)}, layout=layout)
        with pytest.raises(NotImplementedError, match=word):
            hdf5_min.File(p)["images"]
    raw = bytearray(open(p, "rb").read())
    raw[8] = 2                                           # a version-2 superblock (libver='latest')
    open(p, "wb").write(bytes(raw))
    with pytest.raises(NotImplementedError, match="superblock version 2"):
        hdf5_min.File(p)
    open(p, "wb").write(b"not an hdf5 file at all" * 100)
    with pytest.raises(hdf5_min.Hdf5FormatError, match="no HDF5 signature"):
        hdf5_min.File(p)
    with pytest.raises(NotImplementedError):
        hdf5_min.File(p, "w")
