"""Drop-in for the reference package `lib/pointnet2` (module `pointnet2._ext`
and the Python layers built on it)."""
