"""Import-path alias package (reference path utils/)."""
