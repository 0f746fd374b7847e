"""Drop-in replacement for DQO-MAP's `simple_knn` package (submodules/simple-knn) on MI355X: `from simple_knn._C import distCUDA2`."""
