"""Oracle tooling only: minimal stand-in for the third-party `urdf_parser_py`
package (absent from this image) so that /root/reference can be imported in the
development container to generate golden vectors.  Never imported by the product
(torch_robotics_amd has its own URDF reader)."""
