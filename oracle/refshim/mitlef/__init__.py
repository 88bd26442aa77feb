"""CONTAINER-ONLY SHIM: ``mitlef`` (github jdhuang-csm/mittag-leffler) is only *used* by the reference for
Cole-Cole / zga bases (mat1d.py:49-58, basis.py:592-597), which are outside the hot path."""
