pub mod adler;
